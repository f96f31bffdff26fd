// Error channel + misc host entry points of libserl_hip (C ABI, see include/serl_hip.h).
#include <stdarg.h>

#include "common.h"

static thread_local char g_err[512] = "";

void serl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* serl_last_error(void) { return g_err; }

extern "C" int serl_version(void) { return 1; }

// Process-wide matmul/conv precision (the counterpart of jax.config "jax_default_matmul_precision"):
// 0 = "highest": exact fp32 MFMA; 1 = "bf16_3x": fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs, fp32 accumulate.
static int g_matmul_precision = 1;
extern "C" int serl_set_matmul_precision(int mode) {
  SERL_CHECK_ARG(mode == 0 || mode == 1, "set_matmul_precision: mode must be 0 (highest) or 1 (bf16_3x)");
  g_matmul_precision = mode;
  return SERL_OK;
}
extern "C" int serl_get_matmul_precision(void) { return g_matmul_precision; }

// Device query used by the Python loader to fail loudly on a non-gfx950 device.
extern "C" int serl_device_arch(int device, char* out, int out_len) {
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    serl_set_error("hipGetDeviceProperties(%d): %s", device, hipGetErrorString(e));
    return SERL_ERR_HIP;
  }
  snprintf(out, out_len, "%s", prop.gcnArchName);
  return SERL_OK;
}

// ---- host-side JAX-compatible key plumbing (same threefry code as the device kernels) ----
// jax.random.split(key, num) -> out[2*num]
extern "C" int serl_prng_split(unsigned k0, unsigned k1, int num, unsigned* out) {
  SERL_CHECK_ARG(out && num >= 1, "prng_split: bad args");
  for (int i = 0; i < 2 * num; ++i) out[i] = threefry_bits_at(k0, k1, (uint32_t)i, (uint32_t)(2 * num));
  return SERL_OK;
}
// jax.random.fold_in(key, data) = threefry_2x32(key, [0, data])
extern "C" int serl_prng_fold_in(unsigned k0, unsigned k1, unsigned data, unsigned* out) {
  SERL_CHECK_ARG(out, "prng_fold_in: bad args");
  uint32_t x0 = 0, x1 = data;
  threefry2x32(k0, k1, x0, x1);
  out[0] = x0; out[1] = x1;
  return SERL_OK;
}
// jax.random.bits(key, (n,), uint32)
extern "C" int serl_prng_bits(unsigned k0, unsigned k1, int n, unsigned* out) {
  SERL_CHECK_ARG(out && n >= 1, "prng_bits: bad args");
  for (int i = 0; i < n; ++i) out[i] = threefry_bits_at(k0, k1, (uint32_t)i, (uint32_t)n);
  return SERL_OK;
}
// jax.random.choice(key, arange(n), (m,), replace=False) = permutation(key, arange(n))[:m]:
// rounds of {key, sub = split(key); stable sort by random_bits(sub, (n,))}  (serl/networks/ensemble.py:28)
extern "C" int serl_prng_choice(unsigned k0, unsigned k1, int n, int m, int* out) {
  SERL_CHECK_ARG(out && n >= 1 && m >= 1 && m <= n && n <= 4096, "prng_choice: bad args");
  int x[4096];
  for (int i = 0; i < n; ++i) x[i] = i;
  int rounds = 1;  // ceil(3 ln(n) / ln(2^32-1)): 1 for n <= 1625, 2 up to 4096^... (n <= 4096 -> <= 2)
  {
    double r = 3.0 * __builtin_log((double)(n > 1 ? n : 1)) / __builtin_log(4294967295.0);
    rounds = (int)__builtin_ceil(r);
  }
  for (int rd = 0; rd < rounds; ++rd) {
    unsigned nk0 = threefry_bits_at(k0, k1, 0, 4), nk1 = threefry_bits_at(k0, k1, 1, 4);
    unsigned s0 = threefry_bits_at(k0, k1, 2, 4), s1 = threefry_bits_at(k0, k1, 3, 4);
    k0 = nk0; k1 = nk1;
    unsigned keys[4096];
    for (int i = 0; i < n; ++i) keys[i] = threefry_bits_at(s0, s1, (uint32_t)i, (uint32_t)n);
    // stable insertion sort of (keys, x)
    for (int i = 1; i < n; ++i) {
      unsigned kk = keys[i]; int xv = x[i]; int j = i - 1;
      while (j >= 0 && keys[j] > kk) { keys[j + 1] = keys[j]; x[j + 1] = x[j]; --j; }
      keys[j + 1] = kk; x[j + 1] = xv;
    }
  }
  for (int i = 0; i < m; ++i) out[i] = x[i];
  return SERL_OK;
}
