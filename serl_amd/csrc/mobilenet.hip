// Frozen MobileNetV3-Large feature extractor (inference only) behind TwoMobileNetEncoder / MobileNetEncoder
// (serl/networks/encoders/two_mobilenet_encoder.py:19-34, mobilenet_encoder.py:29-48).  The reference gets the network
// from the un-vendored `jeffnet` package ('tf_mobilenetv3_large_100', pretrained); what is implemented is the published
// architecture of that model name with BatchNorm folded into the convolutions, TensorFlow SAME padding, cut after the
// last 1x1 convolution (960 channels, stride 32) -- restated on the CPU in oracle/mobilenet.py (parity unpinned for
// the backbone itself: DESIGN.md).
//
// All activations are NHWC fp32.  Per block:  1x1 expand (serl_gemm, bias + ReLU / hard-swish epilogue) -> depthwise
// k x k (this file: one thread per output pixel x 4 channels, float4, HBM-bound) -> squeeze-excite (mean over the
// pixels, two tiny GEMMs with ReLU / hard-sigmoid epilogues, channel scaling) -> 1x1 project (serl_gemm; the residual
// is added by accumulating into the block's input buffer).
#include <algorithm>

#include "../../include/serl_hip.h"
#include "common.h"
#include "igemm_bf3.h"

void serl_gemm_set_big_n(int n);     // dense.hip: column threshold of serl_gemm's 128 x 128 tiles for the calling thread (0: default)

namespace {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_HSWISH = 2, ACT_HSIGMOID = 3 };

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_HSWISH) return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
  if (act == ACT_HSIGMOID) return fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
  return v;
}

struct Blk { int ir, k, s, exp, cout, se, act; };
constexpr int NBLK = 15, STEM_CH = 16, HEAD_CH = 960;
const Blk BLOCKS[NBLK] = {
    {0, 3, 1, 16, 16, 0, ACT_RELU},
    {1, 3, 2, 64, 24, 0, ACT_RELU},     {1, 3, 1, 72, 24, 0, ACT_RELU},
    {1, 5, 2, 72, 40, 1, ACT_RELU},     {1, 5, 1, 120, 40, 1, ACT_RELU},    {1, 5, 1, 120, 40, 1, ACT_RELU},
    {1, 3, 2, 240, 80, 0, ACT_HSWISH},  {1, 3, 1, 200, 80, 0, ACT_HSWISH},  {1, 3, 1, 184, 80, 0, ACT_HSWISH},
    {1, 3, 1, 184, 80, 0, ACT_HSWISH},  {1, 3, 1, 480, 112, 1, ACT_HSWISH}, {1, 3, 1, 672, 112, 1, ACT_HSWISH},
    {1, 5, 2, 672, 160, 1, ACT_HSWISH}, {1, 5, 1, 960, 160, 1, ACT_HSWISH}, {1, 5, 1, 960, 160, 1, ACT_HSWISH},
};

int make_divisible(double v, int d) {
  int nv = (int)(v + d / 2.0) / d * d;
  if (nv < d) nv = d;
  if (nv < 0.9 * v) nv += d;
  return nv;
}
int se_channels(int exp) { return make_divisible(exp * 0.25, 8); }
void same_pad(int size, int k, int s, int* before, int* out) {   // TensorFlow "SAME"
  *out = (size + s - 1) / s;
  int total = (*out - 1) * s + k - size;
  if (total < 0) total = 0;
  *before = total / 2;
}

// ---- stem: ImageNet normalisation + 3x3 stride-2 convolution (3 -> 16) + bias + hard-swish
// img (n,H,W,3) uint8 (scaled by 1/255) or float already in [0,1]; one thread per output pixel, all 16 channels.
struct StemArgs {
  const void* img; const float* w; const float* b; float* y;
  int n, H, W, OH, OW, pt, pl, in_u8;
};
__global__ __launch_bounds__(256) void stem_kernel(const StemArgs a) {
  // the 432 weights are read with uniform indices straight from memory: scalar loads, an SGPR operand per FMA (from an
  // LDS copy every FMA cost a broadcast ds_read: 184 us per 1024 images against 268 MB of output)
  const float* __restrict__ sw = a.w;
  const float* __restrict__ sb = a.b;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)a.n * a.OH * a.OW;
  if (idx >= total) return;
  const int ox = (int)(idx % a.OW), oy = (int)((idx / a.OW) % a.OH);
  const long long img = idx / ((long long)a.OW * a.OH);
  const float mean[3] = {0.485f, 0.456f, 0.406f}, std[3] = {0.229f, 0.224f, 0.225f};
  float acc[STEM_CH];
#pragma unroll
  for (int f = 0; f < STEM_CH; ++f) acc[f] = sb[f];
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * 2 + ky - a.pt;
    if (iy < 0 || iy >= a.H) continue;   // zero padding applies to the NORMALISED image
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * 2 + kx - a.pl;
      if (ix < 0 || ix >= a.W) continue;
      const long long p = ((img * a.H + iy) * a.W + ix) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = a.in_u8 ? (float)((const unsigned char*)a.img)[p + c] / 255.0f : ((const float*)a.img)[p + c];
        v = (v - mean[c]) / std[c];
        const float* wr = sw + ((ky * 3 + kx) * 3 + c) * STEM_CH;
#pragma unroll
        for (int f = 0; f < STEM_CH; ++f) acc[f] = fmaf(v, wr[f], acc[f]);
      }
    }
  }
  float4* out = (float4*)(a.y + idx * STEM_CH);
#pragma unroll
  for (int f = 0; f < STEM_CH; f += 4)
    out[f / 4] = make_float4(act_apply(acc[f], ACT_HSWISH), act_apply(acc[f + 1], ACT_HSWISH),
                             act_apply(acc[f + 2], ACT_HSWISH), act_apply(acc[f + 3], ACT_HSWISH));
}

// ---- depthwise k x k convolution + bias + activation, NHWC, SAME padding; thread = (output pixel, 4 channels)
struct DwArgs {
  const float* x; const float* w; const float* b; float* y;
  int n, H, W, C4, OH, OW, k, s, pt, pl, act;
};
__global__ __launch_bounds__(256) void dw_kernel(const DwArgs a) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)a.n * a.OH * a.OW * a.C4;
  if (idx >= total) return;
  const int c4 = (int)(idx % a.C4);
  long long p = idx / a.C4;
  const int ox = (int)(p % a.OW); p /= a.OW;
  const int oy = (int)(p % a.OH);
  const long long img = p / a.OH;
  const float4* x = (const float4*)a.x;
  const float4* w = (const float4*)a.w;
  float4 acc = ((const float4*)a.b)[c4];
  for (int ky = 0; ky < a.k; ++ky) {
    const int iy = oy * a.s + ky - a.pt;
    if (iy < 0 || iy >= a.H) continue;
    for (int kx = 0; kx < a.k; ++kx) {
      const int ix = ox * a.s + kx - a.pl;
      if (ix < 0 || ix >= a.W) continue;
      const float4 v = x[((img * a.H + iy) * a.W + ix) * a.C4 + c4];
      const float4 ww = w[(ky * a.k + kx) * a.C4 + c4];
      acc.x = fmaf(v.x, ww.x, acc.x); acc.y = fmaf(v.y, ww.y, acc.y);
      acc.z = fmaf(v.z, ww.z, acc.z); acc.w = fmaf(v.w, ww.w, acc.w);
    }
  }
  ((float4*)a.y)[idx] = make_float4(act_apply(acc.x, a.act), act_apply(acc.y, a.act), act_apply(acc.z, a.act),
                                    act_apply(acc.w, a.act));
}

// ---- squeeze: out[img][c] = mean over the HW pixels; block = (image, 64 channels), 4 pixel phases x 64 channels
__global__ __launch_bounds__(256) void se_pool_kernel(const float* __restrict__ x, float* __restrict__ out, int HW, int C) {
  __shared__ float red[4][64];
  const int img = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  float s = 0.f;
  if (c < C) {
    // 16 loads in flight per thread (the adds keep their order: same bits); a rolled loop was one round trip per pixel
    const float* px = x + (long long)img * HW * C + c;
    int p = ph;
    for (; p + 60 < HW; p += 64) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = px[(long long)(p + 4 * j) * C];
#pragma unroll
      for (int j = 0; j < 16; ++j) s += v[j];
    }
    for (; p < HW; p += 4) s += px[(long long)p * C];
  }
  red[ph][threadIdx.x & 63] = s;
  __syncthreads();
  if (ph == 0 && c < C) out[(long long)img * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)HW;
}
// ---- excite: x[img][p][c] *= gate[img][c], float4
__global__ __launch_bounds__(256) void se_scale_kernel(float* x, const float* gate, long long total4, int HW, int C4) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total4) return;
  const int c4 = (int)(idx % C4);
  const long long img = idx / ((long long)C4 * HW);
  float4 v = ((float4*)x)[idx];
  const float4 g = ((const float4*)gate)[img * C4 + c4];
  v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
  ((float4*)x)[idx] = v;
}

// ---- streaming pointwise convolution for the wide, shallow layers of the first blocks (M = n H W rows of 16..72 channels
// in, 16..72 out: 25..335 MB of activations against a few KB of weights).  The generic 64 x 64-tile GEMM restages the
// weights per tile and runs these at 0.7-1.9 TB/s.  Here a wave keeps the whole weight matrix as bf16 hi / lo MFMA B
// fragments in registers (KS k16 steps x NTL column tiles), walks 32-row tiles of X with the next tile's rows requested
// before the current tile's MFMAs (a lane reads 8 consecutive floats of its row: whole rows are contiguous, so a wave's
// loads are one contiguous span), splits them into hi / lo in registers (the bf16_3x product lo x hi + hi x lo + hi x hi
// of serl_gemm), and stores finished rows straight from the accumulators (32 consecutive floats of a row per half wave).
template <int KS, int NTL>
__global__ __launch_bounds__(256) void pw_stream_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int M, int K,
                                                        int N, int act, int accumulate) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  // B fragments: lane (n = r, h) holds W[16 s + 8 h + j][32 t + r], j < 8
  bf16x8_t bh[KS][NTL], bl[KS][NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    const int n = 32 * t + r;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      u32x4_t hv, lv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k0 = 16 * s + 8 * h + 2 * e;
        const float a = (k0 < K && n < N) ? w[(size_t)k0 * N + n] : 0.f;
        const float b = (k0 + 1 < K && n < N) ? w[(size_t)(k0 + 1) * N + n] : 0.f;
        uint32_t hh, ll;
        split_pair(a, b, hh, ll);
        hv[e] = hh; lv[e] = ll;
      }
      bh[s][t] = __builtin_bit_cast(bf16x8_t, hv);
      bl[s][t] = __builtin_bit_cast(bf16x8_t, lv);
    }
  }
  const int ntiles = (M + 31) / 32;
  // A rows of a tile: lane (row r, h), step s: floats [16 s + 8 h, + 8) of row 32 tile + r.  Unconditional loads from a
  // clamped address, no mask behind them (a select right behind a load makes hipcc wait for it at once): columns past K
  // meet zero weights, rows past M are not stored -- what is read instead is finite activation data.
  auto load_a = [&](int tile, float4 (&v)[KS][2]) {
    const int m = tile * 32 + r;
    const float* row = x + (size_t)((tile < ntiles && m < M) ? m : 0) * K;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int k0 = 16 * s + 8 * h;
      const float4* p = reinterpret_cast<const float4*>(row + (k0 < K ? k0 : 0));
      v[s][0] = p[0]; v[s][1] = p[1];
    }
  };
  float4 cur[KS][2], nxt[KS][2];
  load_a(wave, cur);
  for (int tile = wave; tile < ntiles; tile += nwaves) {
    load_a(tile + nwaves, nxt);
    f32x16 acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      u32x4_t hv, lv;
      uint32_t hh, ll;
      split_pair(cur[s][0].x, cur[s][0].y, hh, ll); hv[0] = hh; lv[0] = ll;
      split_pair(cur[s][0].z, cur[s][0].w, hh, ll); hv[1] = hh; lv[1] = ll;
      split_pair(cur[s][1].x, cur[s][1].y, hh, ll); hv[2] = hh; lv[2] = ll;
      split_pair(cur[s][1].z, cur[s][1].w, hh, ll); hv[3] = hh; lv[3] = ll;
      const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, hv), al = __builtin_bit_cast(bf16x8_t, lv);
      // The product is computed TRANSPOSED (Y^T = W^T X^T: the weight fragments are the MFMA's A operand, the rows' its B
      // operand -- the same registers, the same products summed over the same k: the same bits), so that a lane ends up
      // with 4 consecutive output channels of its OWN row per accumulator quad: 16-byte stores (and residual loads)
      // instead of 4-byte ones scattered over 16 rows.
#pragma unroll
      for (int t = 0; t < NTL; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[s][t], al, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[s][t], ah, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[s][t], ah, acc[t], 0, 0, 0);
      }
    }
    const int m = tile * 32 + r;
    if (m < M) {
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n0 = 32 * t + 8 * g + 4 * h;          // accumulator elements 4 g .. 4 g + 3: channels n0 .. n0 + 3 (N % 4 == 0)
          if (n0 < N) {
            const float4 b4 = *reinterpret_cast<const float4*>(bias + n0);
            float4* dst = reinterpret_cast<float4*>(y + (size_t)m * N + n0);
            float4 v = make_float4(act_apply(acc[t][4 * g] + b4.x, act), act_apply(acc[t][4 * g + 1] + b4.y, act),
                                   act_apply(acc[t][4 * g + 2] + b4.z, act), act_apply(acc[t][4 * g + 3] + b4.w, act));
            if (accumulate) {
              const float4 o = *dst;
              v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
            }
            *dst = v;
          }
        }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) { cur[s][0] = nxt[s][0]; cur[s][1] = nxt[s][1]; }
  }
}

template <int KS, int NTL>
int pw_stream_launch(const float* x, const float* w, const float* b, float* y, long long M, int K, int N, int act,
                     int accumulate, hipStream_t st) {
  const int ntiles = (int)((M + 31) / 32);
  int blocks = (ntiles + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;       // 8 blocks of 4 waves per CU, each wave walks its tiles with a stride
  hipLaunchKernelGGL((pw_stream_kernel<KS, NTL>), dim3(blocks), dim3(256), 0, st, x, w, b, y, (int)M, K, N, act, accumulate);
  SERL_CHECK_LAUNCH("mbv3 pointwise (streaming)");
  return SERL_OK;
}

// ---- block 0 (no expand: depthwise 3x3 stride 1 + ReLU -> 1x1 project 16 -> 16 + bias + residual) in one kernel.  The two
// launches moved the block's 16-channel tensor (268 MB per 1 024 images) five times (read x, write d, read d, read and
// write x); here it is read once and written once, OUT OF PLACE (a pixel's neighbours are other lanes' outputs).  The
// kernel is pw_stream_kernel<1, 1> with the A operand computed instead of loaded: lane (pixel r, h) runs dw_kernel's tap
// loop for channels 8 h .. 8 h + 7 of its pixel (taps outside the image add fmaf(0, w, acc) = acc; loads unconditional from
// clamped coordinates, the next tile's taps requested before the current tile's arithmetic), the 8 results are split and
// multiplied exactly as pw_stream_kernel does, and the residual is the block input at the same position: same bits as
// dw_kernel + pw_stream_kernel(accumulate).
struct DwPwArgs {
  const float* x; const float* wd; const float* bd; const float* wp; const float* bp; float* y;
  int H, W, pt, pl, M;
  FastDiv d_hw, d_w;
};
__device__ float4 g_mb_zero[2];            // 32 bytes of zeros (never written): what a tap outside the image reads
__global__ __launch_bounds__(256) void dwpw16_kernel(const DwPwArgs a) {
  constexpr int C = 16;
  __shared__ __attribute__((aligned(16))) float sw[10 * C];     // depthwise weights [9][16], then the bias
  for (int i = threadIdx.x; i < 10 * C; i += 256) sw[i] = i < 9 * C ? a.wd[i] : a.bd[i - 9 * C];
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  // The product is computed TRANSPOSED (out^T = Wp^T d^T): the weights are the MFMA's A operand -- lane (n = r, h) holds
  // Wp[8 h + j][r], j < 8, the registers pw_stream_kernel uses as its B fragment -- and the depthwise results the B operand
  // (lane (pixel r, h) holds channels 8 h + j, pw_stream_kernel's A fragment).  The same products summed over the same k:
  // the same bits; but the accumulator now holds, for the lane's OWN pixel, output channels 4 h .. 4 h + 3 and 8 + 4 h .. + 3
  // (q = 0..3, 4..7): two 16-byte stores and two 16-byte residual loads per lane instead of sixteen 4-byte ones.
  bf16x8_t wh, wl;
  {
    u32x4_t hv, lv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k0 = 8 * h + 2 * e;
      const float w0 = r < C ? a.wp[k0 * C + r] : 0.f, w1 = r < C ? a.wp[(k0 + 1) * C + r] : 0.f;
      uint32_t hh, ll;
      split_pair(w0, w1, hh, ll);
      hv[e] = hh; lv[e] = ll;
    }
    wh = __builtin_bit_cast(bf16x8_t, hv);
    wl = __builtin_bit_cast(bf16x8_t, lv);
  }
  const float4 bs0 = *reinterpret_cast<const float4*>(a.bp + 4 * h), bs1 = *reinterpret_cast<const float4*>(a.bp + 8 + 4 * h);
  __syncthreads();
  const int ntiles = (a.M + 31) / 32, HW = a.H * a.W;
  const float* const zero = reinterpret_cast<const float*>(g_mb_zero);
  auto load_taps = [&](int tile, float4 (&v)[9][2]) {
    int m = tile * 32 + r;
    if (tile >= ntiles || m >= a.M) m = 0;                  // (valid memory; never stored)
    const int img = (int)fd_div((uint32_t)m, a.d_hw), rem = m - img * HW, oy = (int)fd_div((uint32_t)rem, a.d_w), ox = rem - oy * a.W;
    const float* ximg = a.x + (size_t)img * HW * C + 8 * h;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = oy + ky - a.pt, ix = ox + kx - a.pl;
        const bool in = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        // a tap outside the image reads zeros (pointer select, no data select): fmaf(0, w, acc) = acc, as dw_kernel's skip
        const float4* p = reinterpret_cast<const float4*>(in ? ximg + ((size_t)iy * a.W + ix) * C : zero);
        v[ky * 3 + kx][0] = p[0]; v[ky * 3 + kx][1] = p[1];
      }
  };
  float4 cur[9][2], nxt[9][2];
  load_taps(wave, cur);
  for (int tile = wave; tile < ntiles; tile += nwaves) {
    load_taps(tile + nwaves, nxt);
    // depthwise conv of this lane's pixel, channels 8 h .. 8 h + 7 (dw_kernel's order: bias, then the taps (ky, kx) ascending)
    float4 d0 = *reinterpret_cast<const float4*>(sw + 9 * C + 8 * h), d1 = *reinterpret_cast<const float4*>(sw + 9 * C + 8 * h + 4);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float4 w0 = *reinterpret_cast<const float4*>(sw + t * C + 8 * h), w1 = *reinterpret_cast<const float4*>(sw + t * C + 8 * h + 4);
      const float4 v0 = cur[t][0], v1 = cur[t][1];
      d0.x = fmaf(v0.x, w0.x, d0.x); d0.y = fmaf(v0.y, w0.y, d0.y); d0.z = fmaf(v0.z, w0.z, d0.z); d0.w = fmaf(v0.w, w0.w, d0.w);
      d1.x = fmaf(v1.x, w1.x, d1.x); d1.y = fmaf(v1.y, w1.y, d1.y); d1.z = fmaf(v1.z, w1.z, d1.z); d1.w = fmaf(v1.w, w1.w, d1.w);
    }
    u32x4_t hv, lv;
    uint32_t hh, ll;
    split_pair(fmaxf(d0.x, 0.f), fmaxf(d0.y, 0.f), hh, ll); hv[0] = hh; lv[0] = ll;
    split_pair(fmaxf(d0.z, 0.f), fmaxf(d0.w, 0.f), hh, ll); hv[1] = hh; lv[1] = ll;
    split_pair(fmaxf(d1.x, 0.f), fmaxf(d1.y, 0.f), hh, ll); hv[2] = hh; lv[2] = ll;
    split_pair(fmaxf(d1.z, 0.f), fmaxf(d1.w, 0.f), hh, ll); hv[3] = hh; lv[3] = ll;
    const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, hv), al = __builtin_bit_cast(bf16x8_t, lv);
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, al, acc, 0, 0, 0);      // (pw_stream_kernel's order: lo x hi, hi x lo, hi x hi)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, ah, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, ah, acc, 0, 0, 0);
    const int m = tile * 32 + r;
    if (m < a.M) {
      const float4* xr = reinterpret_cast<const float4*>(a.x + (size_t)m * C + 4 * h);     // the residual: the block's input
      float4* yr = reinterpret_cast<float4*>(a.y + (size_t)m * C + 4 * h);
      const float4 x0 = xr[0], x1 = xr[2];
      float4 o0, o1;
      o0.x = acc[0] + bs0.x; o0.x += x0.x; o0.y = acc[1] + bs0.y; o0.y += x0.y;
      o0.z = acc[2] + bs0.z; o0.z += x0.z; o0.w = acc[3] + bs0.w; o0.w += x0.w;
      o1.x = acc[4] + bs1.x; o1.x += x1.x; o1.y = acc[5] + bs1.y; o1.y += x1.y;
      o1.z = acc[6] + bs1.z; o1.z += x1.z; o1.w = acc[7] + bs1.w; o1.w += x1.w;
      yr[0] = o0; yr[2] = o1;
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) { cur[t][0] = nxt[t][0]; cur[t][1] = nxt[t][1]; }
  }
}

// ---- 1x1 expand + depthwise k x k in one kernel for the high-resolution blocks: the expanded tensor (4x / 3x the block's
// input, the largest tensors of the network: 1.07 GB per 1024 images in block 1) is never written to memory.  A block
// of 4 waves owns a TO x TO tile of depthwise OUTPUT pixels of one image: (i) its TI x TI input pixels (TI = (TO - 1) S +
// KK, halo included) are expanded on the matrix cores exactly as pw_stream_kernel does (same fragments, same k order: the
// same bits) and written, bias + activation applied, to an LDS tile [pixel][Cexp]; (ii) thread = (output pixel, 4
// channels) runs dw_kernel's loop over the taps on that tile; pixels outside the image hold zeros there (the padding of
// the depthwise conv), which add nothing, where dw_kernel skips the tap.  Same values as the two launches.
template <int K_, int CE_, int KK, int S, int TO>
struct EdCfg {      // K_ input channels, CE_ expanded channels: compile-time, so that every index split is a constant division
  static constexpr int KS = (K_ + 15) / 16, NTL = (CE_ + 31) / 32;
  static constexpr int TI = (TO - 1) * S + KK, NPIX = TI * TI, MT = (NPIX + 31) / 32;
};
struct EdArgs {
  const float* x; const float* we; const float* be; const float* wd; const float* bd; float* y;
  int n, H, W, K, CE, OH, OW, pt, pl, act, tiles_x, tiles_y;
};
template <int K_, int CE_, int KK, int S, int TO, int ACT>
__global__ __launch_bounds__(256) void exp_dw_kernel(const EdArgs a) {
  typedef EdCfg<K_, CE_, KK, S, TO> G;
  constexpr int TI = G::TI, NPIX = G::NPIX, MT = G::MT, KS = G::KS, NTL = G::NTL;
  extern __shared__ __attribute__((aligned(16))) float ed_smem[];
  constexpr int CE = CE_, C4 = CE / 4, K = K_;
  float* const E = ed_smem;                              // [NPIX][CE]
  // depthwise weights [KK * KK][CE], then the bias [CE]: staged in LDS when two blocks per CU still fit with them,
  // else read from global memory (12 KB, L1-resident)
  constexpr bool WLDS = 2 * (NPIX * CE + (KK * KK + 1) * CE) * 4 <= 160 * 1024 || 2 * NPIX * CE * 4 > 160 * 1024;
  float* const Wl = ed_smem + NPIX * CE;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r = lane & 31, h = lane >> 5;
  if (WLDS)
    for (int i = tid; i < (KK * KK + 1) * CE; i += 256) Wl[i] = i < KK * KK * CE ? a.wd[i] : a.bd[i - KK * KK * CE];
  const float* const Wd = WLDS ? Wl : a.wd;
  const float* const Bd = WLDS ? Wl + KK * KK * CE : a.bd;
  // expand weights as B fragments: lane (n = r, h) holds W[16 s + 8 h + j][32 t + r], j < 8 (as pw_stream_kernel)
  bf16x8_t bh[KS][NTL], bl[KS][NTL];
  float bs[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    const int n = 32 * t + r;
    bs[t] = n < CE ? a.be[n] : 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      u32x4_t hv, lv;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k0 = 16 * s + 8 * h + 2 * e;
        const float w0 = (k0 < K && n < CE) ? a.we[(size_t)k0 * CE + n] : 0.f;
        const float w1 = (k0 + 1 < K && n < CE) ? a.we[(size_t)(k0 + 1) * CE + n] : 0.f;
        uint32_t hh, ll;
        split_pair(w0, w1, hh, ll);
        hv[e] = hh; lv[e] = ll;
      }
      bh[s][t] = __builtin_bit_cast(bf16x8_t, hv);
      bl[s][t] = __builtin_bit_cast(bf16x8_t, lv);
    }
  }
  const int ntiles = a.n * a.tiles_y * a.tiles_x;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int img = tile / (a.tiles_y * a.tiles_x), trem = tile - img * (a.tiles_y * a.tiles_x);
    const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
    const int iy0 = ty * TO * S - a.pt, ix0 = tx * TO * S - a.pl;
    const float* ximg = a.x + (size_t)img * a.H * a.W * K;
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + TI <= a.H && ix0 + TI <= a.W;       // block-uniform
    __syncthreads();                                       // the previous tile's depthwise reads (and the weight staging) are done
    // ---- (i) expand: M tile = 32 of the tile's input pixels (row-major TI x TI), wave w takes M tiles w, w + 4, ...; the
    // rows of all of them are requested first (one memory round trip per tile and wave, not one per M tile)
    constexpr int MPW = (MT + 3) / 4;
    float4 av[MPW][KS][2];
#pragma unroll
    for (int j = 0; j < MPW; ++j) {
      const int m = min((wid + 4 * j) * 32 + r, NPIX - 1);
      const int my = m / TI, mx = m - my * TI;
      const int iy = min(max(iy0 + my, 0), a.H - 1), ix = min(max(ix0 + mx, 0), a.W - 1);   // clamped: valid memory, value unused
      const float* row = ximg + ((size_t)iy * a.W + ix) * K;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int k0 = 16 * s + 8 * h;
        const float4* p = reinterpret_cast<const float4*>(row + (k0 < K ? k0 : 0));
        av[j][s][0] = p[0]; av[j][s][1] = p[1];
      }
    }
#pragma unroll
    for (int j = 0; j < MPW; ++j) {
      const int mt = wid + 4 * j;
      if (mt >= MT) break;                                  // wave-uniform
      f32x16 acc[NTL];
#pragma unroll
      for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float4 v0 = av[j][s][0], v1 = av[j][s][1];
        u32x4_t hv, lv;
        uint32_t hh, ll;
        split_pair(v0.x, v0.y, hh, ll); hv[0] = hh; lv[0] = ll;
        split_pair(v0.z, v0.w, hh, ll); hv[1] = hh; lv[1] = ll;
        split_pair(v1.x, v1.y, hh, ll); hv[2] = hh; lv[2] = ll;
        split_pair(v1.z, v1.w, hh, ll); hv[3] = hh; lv[3] = ll;
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, hv), al = __builtin_bit_cast(bf16x8_t, lv);
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[s][t], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[s][t], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[s][t], acc[t], 0, 0, 0);
        }
      }
      // rows past the tile's last pixel exist in the last M tile only (wave-uniform test), columns past CE in the last
      // column tile only (compile-time): everything else is stored without a guard
      float* const e0 = E + (mt * 32 + 4 * h) * CE + r;
      if (!interior) {
        // a tile that touches the image border: the zero padding of the depthwise conv applies to the EXPANDED tensor, so
        // pixels outside the image are stored as zeros (the tap loop below then needs no tests: fmaf(0, w, acc) = acc)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int mm = mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
          const int my = mm / TI, mx = mm - my * TI;
          const bool in = (unsigned)(iy0 + my) < (unsigned)a.H && (unsigned)(ix0 + mx) < (unsigned)a.W;
#pragma unroll
          for (int t = 0; t < NTL; ++t) acc[t][q] = in ? act_apply(acc[t][q] + bs[t], ACT) : 0.f;
        }
#pragma unroll
        for (int t = 0; t < NTL; ++t)
          if ((t + 1) * 32 <= CE || r < CE - t * 32) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (mt * 32 + 32 <= NPIX || mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h < NPIX) e0[((q & 3) + 8 * (q >> 2)) * CE + 32 * t] = acc[t][q];
          }
      } else if (mt * 32 + 32 <= NPIX) {
#pragma unroll
        for (int t = 0; t < NTL; ++t)
          if ((t + 1) * 32 <= CE || r < CE - t * 32) {
#pragma unroll
            for (int q = 0; q < 16; ++q) e0[((q & 3) + 8 * (q >> 2)) * CE + 32 * t] = act_apply(acc[t][q] + bs[t], ACT);
          }
      } else {
#pragma unroll
        for (int t = 0; t < NTL; ++t)
          if ((t + 1) * 32 <= CE || r < CE - t * 32) {
#pragma unroll
            for (int q = 0; q < 16; ++q)
              if (mt * 32 + (q & 3) + 8 * (q >> 2) + 4 * h < NPIX) e0[((q & 3) + 8 * (q >> 2)) * CE + 32 * t] = act_apply(acc[t][q] + bs[t], ACT);
          }
      }
    }
    __syncthreads();
    // ---- (ii) depthwise: thread = (output pixel of the tile, 4 channels), dw_kernel's tap loop on the LDS tile (no
    // per-tap tests: out-of-image pixels hold zeros)
    for (int it = tid; it < TO * TO * C4; it += 256) {
      const int c4 = it % C4, pxy = it / C4;
      const int py = pxy / TO, px = pxy - py * TO;
      const int oy = ty * TO + py, ox = tx * TO + px;
      if (oy >= a.OH || ox >= a.OW) continue;
      float4 acc = reinterpret_cast<const float4*>(Bd)[c4];
      const float4* e0 = reinterpret_cast<const float4*>(E + ((py * S) * TI + px * S) * CE) + c4;
      const float4* w0 = reinterpret_cast<const float4*>(Wd) + c4;
#pragma unroll
      for (int ky = 0; ky < KK; ++ky)
#pragma unroll
        for (int kx = 0; kx < KK; ++kx) {
          const float4 v = e0[(ky * TI + kx) * C4];
          const float4 ww = w0[(ky * KK + kx) * C4];
          acc.x = fmaf(v.x, ww.x, acc.x); acc.y = fmaf(v.y, ww.y, acc.y);
          acc.z = fmaf(v.z, ww.z, acc.z); acc.w = fmaf(v.w, ww.w, acc.w);
        }
      reinterpret_cast<float4*>(a.y + (((size_t)img * a.OH + oy) * a.OW + ox) * CE)[c4] =
          make_float4(act_apply(acc.x, ACT), act_apply(acc.y, ACT), act_apply(acc.z, ACT), act_apply(acc.w, ACT));
    }
  }
}

template <int K_, int CE_, int KK, int S, int TO, int ACT>
int exp_dw_launch(EdArgs a, hipStream_t st) {
  typedef EdCfg<K_, CE_, KK, S, TO> G;
  a.tiles_x = cdiv(a.OW, TO); a.tiles_y = cdiv(a.OH, TO);
  constexpr bool WLDS = 2 * (G::NPIX * CE_ + (KK * KK + 1) * CE_) * 4 <= 160 * 1024 || 2 * G::NPIX * CE_ * 4 > 160 * 1024;
  const int lds = (G::NPIX * CE_ + (WLDS ? (KK * KK + 1) * CE_ : 0)) * 4;
  static bool opted = false;
  if (!opted) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)exp_dw_kernel<K_, CE_, KK, S, TO, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024),
                   "mbv3 expand + depthwise");
    opted = true;
  }
  SERL_CHECK_ARG(lds <= 160 * 1024, "mbv3 expand + depthwise: tile does not fit the LDS");
  const long long ntiles = (long long)a.n * a.tiles_x * a.tiles_y;
  // ~8 tiles per block: the set-up (weight fragments, depthwise weights) is amortised, and the blocks are fine-grained
  // enough that the CUs finish together whatever number of them is resident
  const int blocks = (int)std::min<long long>(ntiles, std::max<long long>(256 * 4, ntiles / 8));
  hipLaunchKernelGGL((exp_dw_kernel<K_, CE_, KK, S, TO, ACT>), dim3(blocks), dim3(256), lds, st, a);
  SERL_CHECK_LAUNCH("mbv3 expand + depthwise");
  return SERL_OK;
}

int g_mbv3_fused = -1;      // serl_mbv3_set_fused (A/B and the bit-identity test); -1: SERL_AMD_MBV3_FUSE, default on
bool mbv3_fused_on() {
  if (g_mbv3_fused < 0) g_mbv3_fused = getenv("SERL_AMD_MBV3_FUSE") == nullptr || strcmp(getenv("SERL_AMD_MBV3_FUSE"), "0") != 0;
  return g_mbv3_fused != 0;
}
// returns 1 when a fused kernel took the expand + depthwise pair of this block, 0 when the caller runs the two launches
int expand_depthwise_fused(const float* x, const float* we, const float* be, const float* wd, const float* bd, float* y, int n,
                           int H, int W, int K, int CE, int OH, int OW, int k, int s, int pt, int pl, int act, hipStream_t st,
                           int* done) {
  *done = 0;
  if (g_mbv3_fused < 0) g_mbv3_fused = getenv("SERL_AMD_MBV3_FUSE") == nullptr || strcmp(getenv("SERL_AMD_MBV3_FUSE"), "0") != 0;
  if (!g_mbv3_fused || serl_get_matmul_precision() != 1 || K % 8 != 0 || CE % 4 != 0 || act == ACT_HSIGMOID) return SERL_OK;
  EdArgs a{x, we, be, wd, bd, y, n, H, W, K, CE, OH, OW, pt, pl, act, 0, 0};
  *done = 1;
  if (act == ACT_RELU) {
    if (K == 16 && CE == 64 && k == 3 && s == 2) return exp_dw_launch<16, 64, 3, 2, 8, ACT_RELU>(a, st);     // block 1
    if (K == 24 && CE == 72 && k == 3 && s == 1) return exp_dw_launch<24, 72, 3, 1, 8, ACT_RELU>(a, st);     // block 2
    if (K == 24 && CE == 72 && k == 5 && s == 2) return exp_dw_launch<24, 72, 5, 2, 4, ACT_RELU>(a, st);     // block 3
    if (K == 40 && CE == 120 && k == 5 && s == 1) return exp_dw_launch<40, 120, 5, 1, 8, ACT_RELU>(a, st);   // blocks 4, 5
  }
  *done = 0;
  return SERL_OK;
}

// pointwise convolution / dense layer: Y (M, N) [+]= act(X (M, K) W (K, N) + bias)
int pointwise(const float* x, const float* w, const float* b, float* y, long long M, int K, int N, int act, int accumulate,
              hipStream_t st) {
  SERL_CHECK_ARG(M * (long long)(K > N ? K : N) < (1ll << 31), "mobilenet: chunk too large for 32-bit tensor offsets");
  // (chosen by K and N only, never by M: a batch computed in chunks must give the bits of the whole batch)
  static const bool stream_ok = getenv("SERL_AMD_MBV3_STREAM") == nullptr || strcmp(getenv("SERL_AMD_MBV3_STREAM"), "0") != 0;   // A/B switch
  if (stream_ok && serl_get_matmul_precision() == 1 && K % 8 == 0 && K <= 80 && N <= 96 && N % 4 == 0 && act != ACT_HSIGMOID &&
      (((uintptr_t)b) & 15) == 0 &&
      (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    const int ks = (K + 15) / 16, ntl = (N + 31) / 32;
#define PW_CASE(KS_, NTL_) if (ks == KS_ && ntl == NTL_) return pw_stream_launch<KS_, NTL_>(x, w, b, y, M, K, N, act, accumulate, st);
    PW_CASE(1, 1) PW_CASE(1, 2) PW_CASE(2, 1) PW_CASE(2, 3) PW_CASE(3, 1) PW_CASE(3, 2) PW_CASE(4, 1) PW_CASE(5, 1) PW_CASE(5, 2)
#undef PW_CASE
  }
  // (128 x 128 tiles from 64 columns on: 4.72 -> 4.50 ms per 1 024-image chain against serl_gemm's default of 512)
  serl_gemm_set_big_n(64);
  const int e = serl_gemm(x, w, y, b, 1, (int)M, N, K, K, 0, 0, K, 1, 0, 0, N, 1, 0, N, 0, 1, 0, act, accumulate, 1, 1, st);
  serl_gemm_set_big_n(0);
  return e;
}

struct Offsets { long long stem_k, stem_b, exp_k[NBLK], exp_b[NBLK], dw_k[NBLK], dw_b[NBLK], ser_k[NBLK], ser_b[NBLK],
                 see_k[NBLK], see_b[NBLK], pr_k[NBLK], pr_b[NBLK], head_k, head_b; };
constexpr int ALIGN = 64;   // floats: leaves are laid out like serl_amd/networks/params.py (256-byte aligned)
long long round_up(long long n) { return (n + ALIGN - 1) / ALIGN * ALIGN; }
long long layout(Offsets* o) {
  long long at = 0;
  auto add = [&](long long n) { long long r = at; at += round_up(n); return r; };
  o->stem_k = add(27 * STEM_CH); o->stem_b = add(STEM_CH);
  int cin = STEM_CH;
  for (int i = 0; i < NBLK; ++i) {
    const Blk& b = BLOCKS[i];
    if (b.ir) { o->exp_k[i] = add((long long)cin * b.exp); o->exp_b[i] = add(b.exp); }
    o->dw_k[i] = add((long long)b.k * b.k * b.exp); o->dw_b[i] = add(b.exp);
    if (b.se) {
      const int r = se_channels(b.exp);
      o->ser_k[i] = add((long long)b.exp * r); o->ser_b[i] = add(r);
      o->see_k[i] = add((long long)r * b.exp); o->see_b[i] = add(b.exp);
    }
    o->pr_k[i] = add((long long)b.exp * b.cout); o->pr_b[i] = add(b.cout);
    cin = b.cout;
  }
  o->head_k = add((long long)cin * HEAD_CH); o->head_b = add(HEAD_CH);
  return at;
}

// per-image workspace (floats): two block input/output buffers, the expanded tensor, the depthwise output, SE vectors
struct Plan { long long x, e, d, se; int fh, fw; };
int plan(int H, int W, Plan* pl) {
  int pb, h, w;
  same_pad(H, 3, 2, &pb, &h);
  same_pad(W, 3, 2, &pb, &w);
  long long mx = (long long)h * w * STEM_CH, me = 0, md = 0;
  int cin = STEM_CH;
  for (int i = 0; i < NBLK; ++i) {
    const Blk& b = BLOCKS[i];
    if (b.ir) me = std::max(me, (long long)h * w * b.exp);
    int oh, ow;
    same_pad(h, b.k, b.s, &pb, &oh);
    same_pad(w, b.k, b.s, &pb, &ow);
    md = std::max(md, (long long)oh * ow * b.exp);
    mx = std::max(mx, (long long)oh * ow * b.cout);
    h = oh; w = ow; cin = b.cout;
  }
  (void)cin;
  pl->x = round_up(mx); pl->e = round_up(me); pl->d = round_up(md); pl->se = round_up(960) * 3;
  pl->fh = h; pl->fw = w;
  return SERL_OK;
}

}  // namespace

extern "C" int serl_mbv3_set_fused(int enable) {
  g_mbv3_fused = enable ? 1 : 0;
  return SERL_OK;
}
extern "C" long long serl_mbv3_param_floats(void) {
  Offsets o;
  return layout(&o);
}
extern "C" int serl_mbv3_feature_shape(int H, int W, int* fh, int* fw, int* fc) {
  SERL_CHECK_ARG(H >= 32 && W >= 32 && fh && fw && fc, "mbv3_feature_shape: images must be at least 32x32");
  Plan pl;
  plan(H, W, &pl);
  *fh = pl.fh; *fw = pl.fw; *fc = HEAD_CH;
  return SERL_OK;
}
extern "C" long long serl_mbv3_workspace_bytes(int n, int H, int W) {
  if (n < 1 || H < 32 || W < 32) return -1;
  Plan pl;
  plan(H, W, &pl);
  return (long long)n * (2 * pl.x + pl.e + pl.d + pl.se) * 4;
}

extern "C" int serl_mbv3_forward(const float* params, const void* images, int in_u8, int n, int H, int W, float* workspace,
                                 long long workspace_bytes, float* features, void* stream) {
  SERL_CHECK_ARG(params && images && workspace && features && n >= 1, "mbv3_forward: bad args");
  SERL_CHECK_ARG(H >= 32 && W >= 32, "mbv3_forward: images must be at least 32x32");
  const long long need = serl_mbv3_workspace_bytes(n, H, W);
  SERL_CHECK_ARG(workspace_bytes >= need, "mbv3_forward: workspace needs %lld bytes, %lld given", need, workspace_bytes);
  SERL_CHECK_ARG((((uintptr_t)params | (uintptr_t)workspace | (uintptr_t)features) & 15) == 0, "mbv3_forward: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  Offsets o;
  layout(&o);
  Plan pl;
  plan(H, W, &pl);
  float* X[2] = {workspace, workspace + (long long)n * pl.x};
  float* E = X[1] + (long long)n * pl.x;
  float* D = E + (long long)n * pl.e;
  float* pooled = D + (long long)n * pl.d;
  float* reduced = pooled + (long long)n * round_up(960);
  float* gate = reduced + (long long)n * round_up(960);
  int h, w, pt, pl_;
  same_pad(H, 3, 2, &pt, &h);
  same_pad(W, 3, 2, &pl_, &w);
  {
    StemArgs a{images, params + o.stem_k, params + o.stem_b, X[0], n, H, W, h, w, pt, pl_, in_u8};
    const long long total = (long long)n * h * w;
    hipLaunchKernelGGL(stem_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, st, a);
    SERL_CHECK_LAUNCH("mbv3 stem");
  }
  int cur = 0, cin = STEM_CH;
  for (int i = 0; i < NBLK; ++i) {
    const Blk& b = BLOCKS[i];
    const long long M = (long long)n * h * w;
    const float* xin = X[cur];
    const float* dwin = xin;
    int oh, ow;
    same_pad(h, b.k, b.s, &pt, &oh);
    same_pad(w, b.k, b.s, &pl_, &ow);
    int fused = 0;
    if (b.ir) {
      if (int e = expand_depthwise_fused(xin, params + o.exp_k[i], params + o.exp_b[i], params + o.dw_k[i], params + o.dw_b[i],
                                         D, n, h, w, cin, b.exp, oh, ow, b.k, b.s, pt, pl_, b.act, st, &fused)) return e;
    }
    if (b.ir && !fused) {
      if (int e = pointwise(xin, params + o.exp_k[i], params + o.exp_b[i], E, M, cin, b.exp, b.act, 0, st)) return e;
      dwin = E;
    }
    // block 0: depthwise + projection + residual in one kernel, out of place (dwpw16_kernel)
    if (!b.ir && b.k == 3 && b.s == 1 && b.exp == 16 && b.cout == 16 && cin == 16 && !b.se && b.act == ACT_RELU &&
        serl_get_matmul_precision() == 1 && mbv3_fused_on() && M < (1ll << 31) / 16) {
      DwPwArgs a{xin, params + o.dw_k[i], params + o.dw_b[i], params + o.pr_k[i], params + o.pr_b[i], X[cur ^ 1], h, w, pt, pl_, (int)M,
                 make_fastdiv((uint32_t)(h * w)), make_fastdiv((uint32_t)w)};
      const int ntiles = (int)((M + 31) / 32);
      int blocks = (ntiles + 3) / 4;
      if (blocks > 256 * 8) blocks = 256 * 8;
      hipLaunchKernelGGL(dwpw16_kernel, dim3(blocks), dim3(256), 0, st, a);
      SERL_CHECK_LAUNCH("mbv3 depthwise + projection");
      cur ^= 1;
      continue;              // (stride 1, 16 -> 16: h, w, cin unchanged)
    }
    if (!fused) {
      DwArgs a{dwin, params + o.dw_k[i], params + o.dw_b[i], D, n, h, w, b.exp / 4, oh, ow, b.k, b.s, pt, pl_, b.act};
      const long long total = (long long)n * oh * ow * (b.exp / 4);
      hipLaunchKernelGGL(dw_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, st, a);
      SERL_CHECK_LAUNCH("mbv3 depthwise");
    }
    const long long Mo = (long long)n * oh * ow;
    if (b.se) {
      const int r = se_channels(b.exp);
      hipLaunchKernelGGL(se_pool_kernel, dim3(n, cdiv(b.exp, 64)), dim3(256), 0, st, D, pooled, oh * ow, b.exp);
      SERL_CHECK_LAUNCH("mbv3 se pool");
      if (int e = pointwise(pooled, params + o.ser_k[i], params + o.ser_b[i], reduced, n, b.exp, r, ACT_RELU, 0, st)) return e;
      if (int e = pointwise(reduced, params + o.see_k[i], params + o.see_b[i], gate, n, r, b.exp, ACT_HSIGMOID, 0, st)) return e;
      const long long total4 = Mo * (b.exp / 4);
      hipLaunchKernelGGL(se_scale_kernel, dim3((unsigned)cdivll(total4, 256)), dim3(256), 0, st, D, gate, total4, oh * ow,
                         b.exp / 4);
      SERL_CHECK_LAUNCH("mbv3 se scale");
    }
    const bool residual = b.s == 1 && cin == b.cout;
    float* xout = residual ? X[cur] : X[cur ^ 1];      // residual: accumulate onto the block's input in place
    if (int e = pointwise(D, params + o.pr_k[i], params + o.pr_b[i], xout, Mo, b.exp, b.cout, ACT_NONE, residual, st)) return e;
    if (!residual) cur ^= 1;
    h = oh; w = ow; cin = b.cout;
  }
  return pointwise(X[cur], params + o.head_k, params + o.head_b, features, (long long)n * h * w, cin, HEAD_CH, ACT_HSWISH, 0, st);
}
