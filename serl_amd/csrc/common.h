// Shared host/device helpers for libserl_hip (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define SERL_OK 0
#define SERL_ERR_ARG 1
#define SERL_ERR_HIP 2

void serl_set_error(const char* fmt, ...);

#define SERL_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      serl_set_error(__VA_ARGS__);           \
      return SERL_ERR_ARG;                   \
    }                                        \
  } while (0)

#define SERL_CHECK_HIP(expr, name)                                          \
  do {                                                                      \
    hipError_t e__ = (expr);                                                \
    if (e__ != hipSuccess) {                                                \
      serl_set_error("%s: %s failed: %s", name, #expr, hipGetErrorString(e__)); \
      return SERL_ERR_HIP;                                                  \
    }                                                                       \
  } while (0)

#define SERL_CHECK_LAUNCH(name)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      serl_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return SERL_ERR_HIP;                                                   \
    }                                                                        \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }

// ---- unsigned division by a run-time constant (Granlund-Montgomery round-up method) ----
// q = (t + ((n - t) >> s1)) >> s2, t = umulhi(n, m). Exact for all 32-bit n.
struct FastDiv {
  uint32_t d, m, s1, s2;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  if (d <= 1) {  // n / 1
    f.m = 0; f.s1 = 0; f.s2 = 0;
    return f;
  }
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;  // l = ceil(log2 d)
  f.m = (uint32_t)(((1ull << 32) * ((1ull << l) - d)) / d + 1);
  f.s1 = 1;
  f.s2 = l - 1;
  return f;
}
__device__ __host__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f) {
#if defined(__HIP_DEVICE_COMPILE__)
  uint32_t t = __umulhi(n, f.m);
#else
  uint32_t t = (uint32_t)(((uint64_t)n * f.m) >> 32);
#endif
  return (t + ((n - t) >> f.s1)) >> f.s2;
}

// ---- threefry2x32 (20 rounds), identical on host and device ----
__device__ __host__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
__device__ __host__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
  x0 += ks[0];
  x1 += ks[1];
#define TF_R(r) { x0 += x1; x1 = rotl32(x1, r); x1 ^= x0; }
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)
  x0 += ks[1]; x1 += ks[2] + 1u;
  TF_R(17) TF_R(29) TF_R(16) TF_R(24)
  x0 += ks[2]; x1 += ks[0] + 2u;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)
  x0 += ks[0]; x1 += ks[1] + 3u;
  TF_R(17) TF_R(29) TF_R(16) TF_R(24)
  x0 += ks[1]; x1 += ks[2] + 4u;
  TF_R(13) TF_R(15) TF_R(26) TF_R(6)
  x0 += ks[2]; x1 += ks[0] + 5u;
#undef TF_R
}

// jax "array form": element i of threefry_2x32(key, iota(n)). Counters are split in halves.
__device__ __host__ __forceinline__ uint32_t threefry_bits_at(uint32_t k0, uint32_t k1, uint32_t i, uint32_t n) {
  const uint32_t npad = n + (n & 1u);
  const uint32_t half = npad >> 1;
  uint32_t x0, x1;
  if (i < half) {
    x0 = i;
    x1 = i + half;
    if (x1 >= n) x1 = 0;  // padded element
    threefry2x32(k0, k1, x0, x1);
    return x0;
  } else {
    x0 = i - half;
    x1 = i;
    threefry2x32(k0, k1, x0, x1);
    return x1;
  }
}

// split(key, num)[child] -> (c0, c1)
__device__ __host__ __forceinline__ void threefry_split_child(uint32_t k0, uint32_t k1, uint32_t num, uint32_t child,
                                                              uint32_t& c0, uint32_t& c1) {
  c0 = threefry_bits_at(k0, k1, 2 * child, 2 * num);
  c1 = threefry_bits_at(k0, k1, 2 * child + 1, 2 * num);
}

// XLA's float32 erf_inv (Giles); jax.random.normal = sqrt(2) * erf_inv(uniform(nextafter(-1,0), 1)).
// Every step is a separately rounded fp32 operation (__fmul_rn / __fadd_rn: no FMA contraction), the evaluation XLA's
// HLO expansion of erf_inv describes and the oracle (oracle/prng.py erfinv_f32) restates op by op, so the normal draws
// agree with the oracle up to the last-place differences of log1pf / sqrtf.
__device__ __forceinline__ float erfinv_f32(float x) {
  float w = -log1pf(-__fmul_rn(x, x));
  float p;
#define EI(c) p = __fadd_rn(c, __fmul_rn(p, w));
  if (w < 5.0f) {
    w = __fadd_rn(w, -2.5f);
    p = 2.81022636e-08f;
    EI(3.43273939e-07f) EI(-3.5233877e-06f) EI(-4.39150654e-06f) EI(0.00021858087f) EI(-0.00125372503f)
    EI(-0.00417768164f) EI(0.246640727f) EI(1.50140941f)
  } else {
    w = __fadd_rn(sqrtf(w), -3.0f);
    p = -0.000200214257f;
    EI(0.000100950558f) EI(0.00134934322f) EI(-0.00367342844f) EI(0.00573950773f) EI(-0.0076224613f)
    EI(0.00943887047f) EI(1.00167406f) EI(2.83297682f)
  }
#undef EI
  return fabsf(x) == 1.0f ? copysignf(INFINITY, x) : __fmul_rn(p, x);
}
__device__ __forceinline__ float jax_normal_from_bits(uint32_t bits) {
  const float lo = -0.99999994f;  // nextafter(-1, 0)
  const float u01 = __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f;
  const float u = fmaxf(lo, __fadd_rn(__fmul_rn(u01, __fadd_rn(1.0f, -lo)), lo));
  return __fmul_rn(1.41421356237309515f, erfinv_f32(u));
}
__device__ __forceinline__ float softplusf(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }


