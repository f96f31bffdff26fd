// Frame-stack gather + DrQ random-shift ("pad by 4, crop back") augmentation, uint8, bit-exact.
//
// Replaces, in one pass over HBM, three reference steps:
//   * sliding_window_view(frames, S+1)[indx - S]   serl/data/memory_efficient_replay_buffer.py:155-160
//   * _unpack (obs = [..., :-1], next = [..., 1:])   serl/utils/commons.py:16-24
//   * batched_random_crop(edge pad 4 + dynamic_slice) serl/utils/augmentations.py:5-11,44-48
// Output layout = reference layout: (n, H, W, 3, S) uint8, i.e. channel index c*S + s after the
// reshape of pixel_multiplexer.py:27.
//
// One block = one sample x {obs,next} x TH output rows.  Phase 1 stages the source rows in LDS with
// 16-byte coalesced loads, phase 2 de-interleaves/shifts pixel by pixel inside LDS, phase 3 streams
// the finished rows out with 16-byte coalesced stores.  Frame rows whose byte length W*3 is not a multiple of 16
// (84x84, 100x100: the usual DrQ sizes) take the same kernel with 4-byte or 1-byte lanes (template VecT).
#include "common.h"

#define GC_TH 8

struct GcArgs {
  const unsigned char* src;   // FRAMES: (cap, H, W, 3)   PACKED: (n, H, W, 3, T)
  const long long* indx;      // FRAMES only: (n) replay indices (window = rows indx-S .. indx)
  const int* off;             // (n, 2) int32 (dy, dx) in [0, 2*pad]
  unsigned char* out;         // (n, H, W, 3*S): out[.., c*S+s] = frame (t0 + s) of the window, shifted
  int n, H, W, S, T, t0, pad, cap;
};

template <bool PACKED, typename VecT>
__global__ __launch_bounds__(256) void gather_crop_kernel(const GcArgs a) {
  constexpr int VB = (int)sizeof(VecT);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int j = blockIdx.x, y0 = blockIdx.y * GC_TH;
  const int S = a.S, W = a.W, H = a.H, T = a.T;
  const int rowb = W * 3;                      // bytes of one source frame row
  const int srow = PACKED ? rowb * T : rowb;   // bytes of one staged source row
  const int nsrc = PACKED ? 1 : S;             // staged planes
  unsigned char* s_src = smem;                                  // [nsrc][TH][srow]
  unsigned char* s_out = smem + nsrc * GC_TH * srow;            // [TH][rowb*S]
  const int dy = a.off[2 * j] - a.pad, dx = a.off[2 * j + 1] - a.pad;
  const int rows = min(GC_TH, H - y0);

  // ---- phase 1: global -> LDS, 16 B per lane
  long long w0 = 0;
  if (!PACKED) {
    w0 = a.indx[j] - (T - 1);
    if (w0 < 0) w0 += a.cap - (T - 1);  // python negative index into the (cap - S)-long window view
  }
  const int vec_per_row = srow / VB;
  const int nvec = nsrc * rows * vec_per_row;
  for (int v = threadIdx.x; v < nvec; v += 256) {
    const int p = v / (rows * vec_per_row);
    const int rem = v - p * rows * vec_per_row;
    const int r = rem / vec_per_row, q = rem - r * vec_per_row;
    int sy = y0 + r + dy;
    sy = sy < 0 ? 0 : (sy >= H ? H - 1 : sy);
    const unsigned char* g;
    if (PACKED) g = a.src + ((long long)j * H + sy) * srow + q * VB;
    else g = a.src + ((w0 + p + a.t0) * H + sy) * (long long)rowb + q * VB;
    *(VecT*)(s_src + (p * GC_TH + r) * srow + q * VB) = *(const VecT*)g;
  }
  __syncthreads();

  // ---- phase 2: per output pixel, pull 3 bytes from each of the S frames, interleave as c*S + s
  const int orow = rowb * S;
  for (int px = threadIdx.x; px < rows * W; px += 256) {
    const int r = px / W, x = px - r * W;
    int sx = x + dx;
    sx = sx < 0 ? 0 : (sx >= W ? W - 1 : sx);
    unsigned char* o = s_out + r * orow + x * 3 * S;
    for (int s = 0; s < S; ++s) {
      const unsigned char* sp;
      if (PACKED) sp = s_src + r * srow + sx * 3 * T + (s + a.t0);
      else sp = s_src + (s * GC_TH + r) * srow + sx * 3;
      const int cs = PACKED ? T : 1;
      o[0 * S + s] = sp[0];
      o[1 * S + s] = sp[cs];
      o[2 * S + s] = sp[2 * cs];
    }
  }
  __syncthreads();

  // ---- phase 3: LDS -> global, 16 B per lane
  unsigned char* out = a.out + ((long long)j * H + y0) * orow;
  const int ovec = rows * orow / VB;
  for (int v = threadIdx.x; v < ovec; v += 256) *(VecT*)(out + v * VB) = *(const VecT*)(s_out + v * VB);
}

static int gc_launch(const GcArgs& a, bool packed, hipStream_t st) {
  const int rowb = a.W * 3;
  SERL_CHECK_ARG(a.n >= 1 && a.S >= 1 && a.t0 >= 0 && a.t0 + a.S <= a.T, "gather_crop: frames t0..t0+S-1 outside the window");
  SERL_CHECK_ARG(a.H >= 1 && a.W >= 1, "gather_crop: bad image size");
  // lane width: every staged row, output row and tile start is a multiple of W*3 bytes from the buffer start
  const uintptr_t al = (uintptr_t)a.src | (uintptr_t)a.out | (uintptr_t)rowb;
  const int vb = (al & 15) == 0 ? 16 : (al & 3) == 0 ? 4 : 1;
  const int srow = packed ? rowb * a.T : rowb;
  const int nsrc = packed ? 1 : a.S;
  const size_t lds = (size_t)nsrc * GC_TH * srow + (size_t)GC_TH * rowb * a.S;
  SERL_CHECK_ARG(lds <= 64 * 1024, "gather_crop: tile needs %zu B of LDS", lds);
  dim3 grid(a.n, cdiv(a.H, GC_TH));
#define GC_LAUNCH(P, V) hipLaunchKernelGGL((gather_crop_kernel<P, V>), grid, dim3(256), lds, st, a)
  if (packed) {
    if (vb == 16) GC_LAUNCH(true, uint4); else if (vb == 4) GC_LAUNCH(true, uint32_t); else GC_LAUNCH(true, unsigned char);
  } else {
    if (vb == 16) GC_LAUNCH(false, uint4); else if (vb == 4) GC_LAUNCH(false, uint32_t); else GC_LAUNCH(false, unsigned char);
  }
#undef GC_LAUNCH
  SERL_CHECK_LAUNCH("gather_crop");
  return SERL_OK;
}

// frames: device-resident frame store (cap,H,W,3); indx: device int64 (n); the window of sample j is
// store rows indx[j]-S_win .. indx[j] (S_win+1 = T frames); output takes frames t0 .. t0+S-1 of it.
extern "C" int serl_gather_crop_frames(const unsigned char* frames, long long cap, const long long* indx, int T, int t0,
                                       const int* off, unsigned char* out, int n, int H, int W, int S, int pad,
                                       void* stream) {
  SERL_CHECK_ARG(frames && indx && off && out, "gather_crop_frames: null pointer");
  SERL_CHECK_ARG(cap > T && cap < (1ll << 31), "gather_crop_frames: bad capacity");
  GcArgs a{frames, indx, off, out, n, H, W, S, T, t0, pad, (int)cap};
  return gc_launch(a, false, (hipStream_t)stream);
}

// src: (n,H,W,3,T) stacked frames (T = S+1 for sample(pack_obs_and_next_obs=True), T = S when unpacked).
extern "C" int serl_gather_crop_packed(const unsigned char* src, int T, int t0, const int* off, unsigned char* out,
                                       int n, int H, int W, int S, int pad, void* stream) {
  SERL_CHECK_ARG(src && off && out, "gather_crop_packed: null pointer");
  GcArgs a{src, nullptr, off, out, n, H, W, S, T, t0, pad, 0};
  return gc_launch(a, true, (hipStream_t)stream);
}

// Plain window gather (no shift), used by the device-resident buffer's sample(): out (n,H,W,3,S+1).
__global__ __launch_bounds__(256) void gather_window_kernel(const unsigned char* frames, int cap, const long long* indx,
                                                            unsigned char* out, int HW3, int S) {
  const int j = blockIdx.x;
  long long w0 = indx[j] - S;
  if (w0 < 0) w0 += cap - S;
  const int T = S + 1;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < HW3; i += gridDim.y * 256) {
    for (int t = 0; t < T; ++t) out[((long long)j * HW3 + i) * T + t] = frames[(w0 + t) * HW3 + i];
  }
}
extern "C" int serl_gather_window(const unsigned char* frames, long long cap, const long long* indx, unsigned char* out,
                                  int n, int HW3, int S, void* stream) {
  SERL_CHECK_ARG(frames && indx && out && n >= 1 && HW3 >= 1 && S >= 1 && cap > S, "gather_window: bad args");
  hipLaunchKernelGGL(gather_window_kernel, dim3(n, 16), dim3(256), 0, (hipStream_t)stream, frames, (int)cap, indx, out,
                     HW3, S);
  SERL_CHECK_LAUNCH("gather_window");
  return SERL_OK;
}

// crop offsets: keys = split(key, n_total)[row]; (dy, dx) = randint(keys[row], (2,), 0, 2*pad+1)
// (jax.random.randint: two splits, hi/lo 32-bit words, ((hi % span) * (2^32 % span) + lo % span) % span).
__global__ void crop_offsets_kernel(uint32_t k0, uint32_t k1, int n, int row_offset, int n_total, int span, int* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t s0, s1;
  threefry_split_child(k0, k1, (uint32_t)n_total, (uint32_t)(row_offset + i), s0, s1);
  uint32_t a0, a1, b0, b1;
  threefry_split_child(s0, s1, 2, 0, a0, a1);   // k1 of randint
  threefry_split_child(s0, s1, 2, 1, b0, b1);   // k2
  uint32_t h0 = 0, h1 = 1, l0 = 0, l1 = 1;      // random_bits(k, (2,)) = threefry(k, (0, 1))
  threefry2x32(a0, a1, h0, h1);
  threefry2x32(b0, b1, l0, l1);
  const uint32_t sp = (uint32_t)span;
  uint32_t mult = 65536u % sp;
  mult = (mult * mult) % sp;
  out[2 * i] = (int)(((h0 % sp) * mult + (l0 % sp)) % sp);
  out[2 * i + 1] = (int)(((h1 % sp) * mult + (l1 % sp)) % sp);
}
extern "C" int serl_crop_offsets(unsigned k0, unsigned k1, int n, int row_offset, int n_total, int pad, int* out,
                                 void* stream) {
  SERL_CHECK_ARG(out && n >= 1 && pad >= 0 && row_offset >= 0 && row_offset + n <= n_total, "crop_offsets: bad args");
  hipLaunchKernelGGL(crop_offsets_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, k0, k1, n, row_offset,
                     n_total, 2 * pad + 1, out);
  SERL_CHECK_LAUNCH("crop_offsets");
  return SERL_OK;
}

// Device-side index draw of Dataset.sample_jax: indx = jax.random.randint(key, (n,), minval, maxval) plus indx.max().
// random_bits(k, (n,)) = threefry_2x32(k, arange(n)): counts are cut in halves (odd n padded with one 0), pair i is
// (count i, count i + half) and yields elements i and i + half.
__global__ void randint_kernel(uint32_t k0, uint32_t k1, int n, int minval, uint32_t span, int* out, int* out_max) {
  const int half = (n + 1) / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= half) return;
  uint32_t a0, a1, b0, b1;
  threefry_split_child(k0, k1, 2, 0, a0, a1);
  threefry_split_child(k0, k1, 2, 1, b0, b1);
  const bool pad = i + half >= n;
  uint32_t h0 = (uint32_t)i, h1 = pad ? 0u : (uint32_t)(i + half), l0 = h0, l1 = h1;
  threefry2x32(a0, a1, h0, h1);
  threefry2x32(b0, b1, l0, l1);
  uint32_t mult = 65536u % span;
  mult = (mult * mult) % span;
  const int v0 = minval + (int)(((h0 % span) * mult + (l0 % span)) % span);
  out[i] = v0;
  int mx = v0;
  if (!pad) {
    const int v1 = minval + (int)(((h1 % span) * mult + (l1 % span)) % span);
    out[i + half] = v1;
    mx = max(mx, v1);
  }
  if (out_max) atomicMax(out_max, mx);
}
extern "C" int serl_prng_randint(unsigned k0, unsigned k1, int n, int minval, int maxval, int* out, int* out_max,
                                 void* stream) {
  SERL_CHECK_ARG(out && n >= 1, "prng_randint: bad args");
  hipStream_t st = (hipStream_t)stream;
  const uint32_t span = maxval > minval ? (uint32_t)((long long)maxval - minval) : 1u;
  if (out_max) SERL_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)out_max, (int)0x80000000, 1, st), "prng_randint");
  hipLaunchKernelGGL(randint_kernel, dim3(cdiv((n + 1) / 2, 256)), dim3(256), 0, st, k0, k1, n, minval, span, out,
                     out_max);
  SERL_CHECK_LAUNCH("prng_randint");
  return SERL_OK;
}

// jnp.take(src, indx, axis=0) on rows of row_bytes bytes (sample_jax, dataset.py:122-124): out[j] = src[indx[j]].
template <typename T>
__global__ __launch_bounds__(256) void take_rows_kernel(const T* src, long long nrows, long long row_elems,
                                                        const int* indx, T* out) {
  const int j = blockIdx.x;
  long long r = indx[j];
  r = r < 0 ? 0 : (r >= nrows ? nrows - 1 : r);
  const T* s = src + r * row_elems;
  T* d = out + (long long)j * row_elems;
  for (long long i = blockIdx.y * 256 + threadIdx.x; i < row_elems; i += (long long)gridDim.y * 256) d[i] = s[i];
}
extern "C" int serl_take_rows(const void* src, long long nrows, long long row_bytes, const int* indx, void* out, int n,
                              void* stream) {
  SERL_CHECK_ARG(src && indx && out && n >= 1 && nrows >= 1 && row_bytes >= 1, "take_rows: bad args");
  hipStream_t st = (hipStream_t)stream;
  const bool v16 = row_bytes % 16 == 0 && (((uintptr_t)src | (uintptr_t)out) & 15) == 0;
  const bool v4 = row_bytes % 4 == 0 && (((uintptr_t)src | (uintptr_t)out) & 3) == 0;
  const long long elems = v16 ? row_bytes / 16 : v4 ? row_bytes / 4 : row_bytes;
  const long long gyl = cdivll(elems, 256);
  const int gy = (int)(gyl < 64 ? gyl : 64);
  if (v16) hipLaunchKernelGGL(take_rows_kernel<uint4>, dim3(n, gy), dim3(256), 0, st, (const uint4*)src, nrows, elems, indx, (uint4*)out);
  else if (v4) hipLaunchKernelGGL(take_rows_kernel<uint32_t>, dim3(n, gy), dim3(256), 0, st, (const uint32_t*)src, nrows, elems, indx, (uint32_t*)out);
  else hipLaunchKernelGGL(take_rows_kernel<unsigned char>, dim3(n, gy), dim3(256), 0, st, (const unsigned char*)src, nrows, elems, indx, (unsigned char*)out);
  SERL_CHECK_LAUNCH("take_rows");
  return SERL_OK;
}
