// SpatialLearnedEmbeddings (serl/networks/spatial.py:13-27):
//   out[b, c*F + f] = sum_{hw} feat[b,hw,c] * kernel[hw,c,f]        (F = 8 learned embeddings / channel)
// ~0.1 MMAC per image: VALU kernels with channel-contiguous (coalesced) access.
#include "common.h"
#include "igemm_bf3.h"   // split_pair

// feat / dfeat are fp32 [B][HW][C] or, with SPLIT, S format (bf16 hi / lo planes per pixel, sgemm.h)
template <bool SPLIT>
__device__ __forceinline__ float feat_at(const void* feat, long long pix, int C, int c) {
  if constexpr (SPLIT) {
    const unsigned short* f = (const unsigned short*)feat + pix * 2 * C + c;
    return __uint_as_float((uint32_t)f[0] << 16) + __uint_as_float((uint32_t)f[C] << 16);
  } else {
    return ((const float*)feat)[pix * C + c];
  }
}

#define SLE_F 8
#define SLE_MAXG 8
struct SleFwdArgs { const void* feat[SLE_MAXG]; const float* ker[SLE_MAXG]; float* out[SLE_MAXG]; };
struct SleBwdArgs { const float* dout[SLE_MAXG]; const float* ker[SLE_MAXG]; const void* feat[SLE_MAXG]; void* dfeat[SLE_MAXG]; float* dker[SLE_MAXG];
                    int nomask; };   // nomask: the features are not a ReLU output (frozen-backbone embeddings): plain dfeat

// thread = (b, c); grid.y splits hw into chunks whose partials are summed in a fixed order by the caller
// (nchunk == 1 for the 7x7 maps of the stride-2 encoder).
template <bool SPLIT>
__global__ __launch_bounds__(256) void sle_fwd_kernel(const SleFwdArgs a, int ldo, int B, int HW, int C) {
  const void* __restrict__ feat = a.feat[blockIdx.y];
  const float* __restrict__ ker = a.ker[blockIdx.y];
  float* __restrict__ out = a.out[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float acc[SLE_F];
#pragma unroll
  for (int f = 0; f < SLE_F; ++f) acc[f] = 0.f;
  const float4* kp = (const float4*)(ker + (long long)c * SLE_F);
#pragma unroll 7
  for (int hw = 0; hw < HW; ++hw) {
    const float x = feat_at<SPLIT>(feat, (long long)b * HW + hw, C, c);
    const float4 k0 = kp[(long long)hw * C * 2], k1 = kp[(long long)hw * C * 2 + 1];
    acc[0] += x * k0.x; acc[1] += x * k0.y; acc[2] += x * k0.z; acc[3] += x * k0.w;
    acc[4] += x * k1.x; acc[5] += x * k1.y; acc[6] += x * k1.z; acc[7] += x * k1.w;
  }
  float* op = out + (long long)b * ldo + c * SLE_F;
#pragma unroll
  for (int f = 0; f < SLE_F; ++f) op[f] = acc[f];
}

// Variant for the bench shapes: block = 32 channels x 8 slices of hw, for SLE_NB images.  The kernel[hw, c, :] row is
// loaded once per SLE_NB images (one image per thread makes every block re-read the whole 400 KB kernel from L2,
// 8x the feature bytes) and the hw loop is 8x shorter; the 8 slice partials are summed through LDS in a fixed order.
#define SLE_NB 4
template <bool SPLIT>
__global__ __launch_bounds__(256) void sle_fwd_sliced_kernel(const SleFwdArgs a, int ldo, int B, int HW, int C) {
  __shared__ float part[8][SLE_NB * 32 * SLE_F + 8];
  const void* __restrict__ feat = a.feat[blockIdx.z];
  const float* __restrict__ ker = a.ker[blockIdx.z];
  float* __restrict__ out = a.out[blockIdx.z];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, b0 = blockIdx.y * SLE_NB;
  const int per = (HW + 7) / 8;
  const int hw0 = sl * per, hw1 = min(HW, hw0 + per);
  float acc[SLE_NB][SLE_F];
#pragma unroll
  for (int j = 0; j < SLE_NB; ++j)
#pragma unroll
    for (int f = 0; f < SLE_F; ++f) acc[j][f] = 0.f;
  if (c < C) {
    const float4* kp = (const float4*)(ker + (long long)c * SLE_F);
    for (int hw = hw0; hw < hw1; ++hw) {
      const float4 k0 = kp[(long long)hw * C * 2], k1 = kp[(long long)hw * C * 2 + 1];
#pragma unroll
      for (int j = 0; j < SLE_NB; ++j) {
        const int b = b0 + j < B ? b0 + j : B - 1;          // clamped: tail images are computed and not stored
        const float x = feat_at<SPLIT>(feat, (long long)b * HW + hw, C, c);
        acc[j][0] += x * k0.x; acc[j][1] += x * k0.y; acc[j][2] += x * k0.z; acc[j][3] += x * k0.w;
        acc[j][4] += x * k1.x; acc[j][5] += x * k1.y; acc[j][6] += x * k1.z; acc[j][7] += x * k1.w;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < SLE_NB; ++j)
#pragma unroll
    for (int f = 0; f < SLE_F; ++f) part[sl][(j * 32 + cl) * SLE_F + f] = acc[j][f];
  __syncthreads();
  // SLE_NB * 32 * 8 = 1024 outputs per block, 4 per thread: (image j, channel cl2, features f4..f4+3)
  for (int o = threadIdx.x; o < SLE_NB * 32 * SLE_F / 4; o += 256) {
    const int j = o / (32 * SLE_F / 4), rem = o - j * (32 * SLE_F / 4);      // rem = cl2 * 2 + half
    const int cc = blockIdx.x * 32 + (rem >> 1);
    if (b0 + j < B && cc < C) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const float4 p4 = *reinterpret_cast<const float4*>(&part[s8][(j * 32 * SLE_F) + rem * 4]);
        v.x += p4.x; v.y += p4.y; v.z += p4.z; v.w += p4.w;
      }
      *reinterpret_cast<float4*>(out + (long long)(b0 + j) * ldo + (long long)blockIdx.x * 32 * SLE_F + rem * 4) = v;
    }
  }
}

extern "C" int serl_sle_fwd(int ngroups, const void* const* feat, const float* const* ker, float* const* out, int ldo,
                            int B, int HW, int C, int F, int feat_split, void* stream) {
  SERL_CHECK_ARG(ngroups >= 1 && ngroups <= SLE_MAXG, "sle_fwd: ngroups must be in [1,%d]", SLE_MAXG);
  SERL_CHECK_ARG(feat && ker && out && B >= 1 && HW >= 1 && C >= 1, "sle_fwd: bad args");
  SERL_CHECK_ARG(F == SLE_F, "sle_fwd: num_features must be %d", SLE_F);
  SleFwdArgs a;
  for (int i = 0; i < ngroups; ++i) {
    SERL_CHECK_ARG(feat[i] && ker[i] && out[i] && (((uintptr_t)ker[i]) & 15) == 0, "sle_fwd: null or unaligned (16 B) pointer");
    a.feat[i] = feat[i]; a.ker[i] = ker[i]; a.out[i] = out[i];
  }
  bool aligned = (ldo & 3) == 0;
  for (int i = 0; i < ngroups; ++i) aligned = aligned && (((uintptr_t)out[i]) & 15) == 0;
  if (aligned && B >= 64 && HW >= 16) {            // enough work per block: the sliced variant
    const dim3 g2(cdiv(C, 32), cdiv(B, SLE_NB), ngroups);
    if (feat_split)
      hipLaunchKernelGGL(sle_fwd_sliced_kernel<true>, g2, dim3(256), 0, (hipStream_t)stream, a, ldo, B, HW, C);
    else
      hipLaunchKernelGGL(sle_fwd_sliced_kernel<false>, g2, dim3(256), 0, (hipStream_t)stream, a, ldo, B, HW, C);
    SERL_CHECK_LAUNCH("sle_fwd");
    return SERL_OK;
  }
  const dim3 grid(cdiv(B * C, 256), ngroups);
  if (feat_split)
    hipLaunchKernelGGL(sle_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a, ldo, B, HW, C);
  else
    hipLaunchKernelGGL(sle_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a, ldo, B, HW, C);
  SERL_CHECK_LAUNCH("sle_fwd");
  return SERL_OK;
}

// dfeat[b,hw,c] = (feat > 0) * sum_f dout[b, c*F+f] * kernel[hw,c,f]   (ReLU of the last conv folded in)
template <bool SPLIT>
__global__ __launch_bounds__(256) void sle_bwd_data_kernel(const SleBwdArgs a, int ldo, int B, int HW, int C) {
  const float* __restrict__ dout = a.dout[blockIdx.y];
  const float* __restrict__ ker = a.ker[blockIdx.y];
  const void* __restrict__ feat = a.feat[blockIdx.y];
  void* __restrict__ dfeat = a.dfeat[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float d[SLE_F];
  const float* dp = dout + (long long)b * ldo + c * SLE_F;
#pragma unroll
  for (int f = 0; f < SLE_F; ++f) d[f] = dp[f];
  const float4* kp = (const float4*)(ker + (long long)c * SLE_F);
  for (int hw = 0; hw < HW; ++hw) {
    const float4 k0 = kp[(long long)hw * C * 2], k1 = kp[(long long)hw * C * 2 + 1];
    float s = d[0] * k0.x;
    s += d[1] * k0.y; s += d[2] * k0.z; s += d[3] * k0.w;
    s += d[4] * k1.x; s += d[5] * k1.y; s += d[6] * k1.z; s += d[7] * k1.w;
    const long long pix = (long long)b * HW + hw;
    const float v = (a.nomask || feat_at<SPLIT>(feat, pix, C, c) > 0.f) ? s : 0.f;
    if constexpr (SPLIT) {
      uint32_t h, l;
      split_pair(v, 0.f, h, l);
      unsigned short* d = (unsigned short*)dfeat + pix * 2 * C + c;
      d[0] = (unsigned short)h; d[C] = (unsigned short)l;
    } else {
      ((float*)dfeat)[pix * C + c] = v;
    }
  }
}

// Sliced variant (as sle_fwd_sliced_kernel): block = 32 channels x 8 hw slices for SLE_NB images; every (b, hw, c) output
// is independent, the kernel row is loaded once per SLE_NB images and the hw loop is 8x shorter.
template <bool SPLIT>
__global__ __launch_bounds__(256) void sle_bwd_data_sliced_kernel(const SleBwdArgs a, int ldo, int B, int HW, int C) {
  const float* __restrict__ dout = a.dout[blockIdx.z];
  const float* __restrict__ ker = a.ker[blockIdx.z];
  const void* __restrict__ feat = a.feat[blockIdx.z];
  void* __restrict__ dfeat = a.dfeat[blockIdx.z];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl, b0 = blockIdx.y * SLE_NB;
  if (c >= C) return;
  const int per = (HW + 7) / 8;
  const int hw0 = sl * per, hw1 = min(HW, hw0 + per);
  float d[SLE_NB][SLE_F];
#pragma unroll
  for (int j = 0; j < SLE_NB; ++j) {
    const int b = b0 + j < B ? b0 + j : B - 1;
    const float* dp = dout + (long long)b * ldo + c * SLE_F;
#pragma unroll
    for (int f = 0; f < SLE_F; ++f) d[j][f] = dp[f];
  }
  const float4* kp = (const float4*)(ker + (long long)c * SLE_F);
  for (int hw = hw0; hw < hw1; ++hw) {
    const float4 k0 = kp[(long long)hw * C * 2], k1 = kp[(long long)hw * C * 2 + 1];
#pragma unroll
    for (int j = 0; j < SLE_NB; ++j) {
      if (b0 + j >= B) break;
      float s = d[j][0] * k0.x;
      s += d[j][1] * k0.y; s += d[j][2] * k0.z; s += d[j][3] * k0.w;
      s += d[j][4] * k1.x; s += d[j][5] * k1.y; s += d[j][6] * k1.z; s += d[j][7] * k1.w;
      const long long pix = (long long)(b0 + j) * HW + hw;
      const float v = (a.nomask || feat_at<SPLIT>(feat, pix, C, c) > 0.f) ? s : 0.f;
      if constexpr (SPLIT) {
        uint32_t h, l;
        split_pair(v, 0.f, h, l);
        unsigned short* o = (unsigned short*)dfeat + pix * 2 * C + c;
        o[0] = (unsigned short)h; o[C] = (unsigned short)l;
      } else {
        ((float*)dfeat)[pix * C + c] = v;
      }
    }
  }
}

// dker[hw,c,f] = sum_b feat[b,hw,c] * dout[b, c*F+f]; block = 64 channels x 4 batch lanes, fixed-order LDS sum.
template <bool SPLIT>
__global__ __launch_bounds__(256) void sle_bwd_kernel_kernel(const SleBwdArgs a, int ldo, int B, int HW, int C) {
  const float* __restrict__ dout = a.dout[blockIdx.z];
  const void* __restrict__ feat = a.feat[blockIdx.z];
  float* __restrict__ dker = a.dker[blockIdx.z];
  __shared__ float red[4][64][SLE_F + 1];
  const int cl = threadIdx.x & 63, bl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, hw = blockIdx.y;
  float acc[SLE_F];
#pragma unroll
  for (int f = 0; f < SLE_F; ++f) acc[f] = 0.f;
  if (c < C) {
    for (int b = bl; b < B; b += 4) {
      const float x = feat_at<SPLIT>(feat, (long long)b * HW + hw, C, c);
      const float* dp = dout + (long long)b * ldo + c * SLE_F;
#pragma unroll
      for (int f = 0; f < SLE_F; ++f) acc[f] += x * dp[f];
    }
  }
#pragma unroll
  for (int f = 0; f < SLE_F; ++f) red[bl][cl][f] = acc[f];
  __syncthreads();
  if (bl == 0 && c < C) {
    float* op = dker + ((long long)hw * C + c) * SLE_F;
#pragma unroll
    for (int f = 0; f < SLE_F; ++f) op[f] = (red[0][cl][f] + red[1][cl][f]) + (red[2][cl][f] + red[3][cl][f]);
  }
}

extern "C" int serl_sle_bwd(int ngroups, const float* const* dout, int ldo, const float* const* ker,
                            const void* const* feat, void* const* dfeat, float* const* dker, int B, int HW, int C, int F,
                            int feat_split, void* stream) {
  SERL_CHECK_ARG(ngroups >= 1 && ngroups <= SLE_MAXG, "sle_bwd: ngroups must be in [1,%d]", SLE_MAXG);
  SERL_CHECK_ARG(dout && ker && feat && B >= 1 && HW >= 1 && C >= 1, "sle_bwd: bad args");
  SERL_CHECK_ARG(F == SLE_F, "sle_bwd: num_features must be %d", SLE_F);
  hipStream_t st = (hipStream_t)stream;
  SleBwdArgs a;
  a.nomask = (feat_split & 2) != 0;
  feat_split &= 1;
  for (int i = 0; i < ngroups; ++i) {
    SERL_CHECK_ARG(dout[i] && ker[i] && feat[i] && (!dfeat || dfeat[i]) && (!dker || dker[i]), "sle_bwd: null pointer");
    a.dout[i] = dout[i]; a.ker[i] = ker[i]; a.feat[i] = feat[i];
    a.dfeat[i] = dfeat ? dfeat[i] : nullptr; a.dker[i] = dker ? dker[i] : nullptr;
  }
  if (dfeat && B >= 64 && HW >= 16) {
    const dim3 g2(cdiv(C, 32), cdiv(B, SLE_NB), ngroups);
    if (feat_split)
      hipLaunchKernelGGL(sle_bwd_data_sliced_kernel<true>, g2, dim3(256), 0, st, a, ldo, B, HW, C);
    else
      hipLaunchKernelGGL(sle_bwd_data_sliced_kernel<false>, g2, dim3(256), 0, st, a, ldo, B, HW, C);
    SERL_CHECK_LAUNCH("sle_bwd_data");
  } else if (dfeat) {
    const dim3 grid(cdiv(B * C, 256), ngroups);
    if (feat_split)
      hipLaunchKernelGGL(sle_bwd_data_kernel<true>, grid, dim3(256), 0, st, a, ldo, B, HW, C);
    else
      hipLaunchKernelGGL(sle_bwd_data_kernel<false>, grid, dim3(256), 0, st, a, ldo, B, HW, C);
    SERL_CHECK_LAUNCH("sle_bwd_data");
  }
  if (dker) {
    const dim3 grid(cdiv(C, 64), HW, ngroups);
    if (feat_split)
      hipLaunchKernelGGL(sle_bwd_kernel_kernel<true>, grid, dim3(256), 0, st, a, ldo, B, HW, C);
    else
      hipLaunchKernelGGL(sle_bwd_kernel_kernel<false>, grid, dim3(256), 0, st, a, ldo, B, HW, C);
    SERL_CHECK_LAUNCH("sle_bwd_kernel");
  }
  return SERL_OK;
}
