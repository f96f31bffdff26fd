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

// thread = (b, c); grid.y splits hw into chunks whose partials are summed in a fixed order by the caller
// (nchunk == 1 for the 7x7 maps of the stride-2 encoder).
template <bool SPLIT>
__global__ __launch_bounds__(256) void sle_fwd_kernel(const void* __restrict__ feat, const float* __restrict__ ker,
                                                      float* __restrict__ out, int ldo, int B, int HW, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float acc[SLE_F];
#pragma unroll
  for (int f = 0; f < SLE_F; ++f) acc[f] = 0.f;
  const float4* kp = (const float4*)(ker + (long long)c * SLE_F);
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

extern "C" int serl_sle_fwd(const void* feat, const float* ker, float* out, int ldo, int B, int HW, int C, int F,
                            int feat_split, void* stream) {
  SERL_CHECK_ARG(feat && ker && out && B >= 1 && HW >= 1 && C >= 1, "sle_fwd: bad args");
  SERL_CHECK_ARG(F == SLE_F, "sle_fwd: num_features must be %d", SLE_F);
  SERL_CHECK_ARG((((uintptr_t)ker) & 15) == 0, "sle_fwd: kernel must be 16-byte aligned");
  if (feat_split)
    hipLaunchKernelGGL(sle_fwd_kernel<true>, dim3(cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)stream, feat, ker, out, ldo, B, HW, C);
  else
    hipLaunchKernelGGL(sle_fwd_kernel<false>, dim3(cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)stream, feat, ker, out, ldo, B, HW, C);
  SERL_CHECK_LAUNCH("sle_fwd");
  return SERL_OK;
}

// dfeat[b,hw,c] = (feat > 0) * sum_f dout[b, c*F+f] * kernel[hw,c,f]   (ReLU of the last conv folded in)
template <bool SPLIT>
__global__ __launch_bounds__(256) void sle_bwd_data_kernel(const float* __restrict__ dout, int ldo,
                                                           const float* __restrict__ ker,
                                                           const void* __restrict__ feat, void* __restrict__ dfeat,
                                                           int B, int HW, int C) {
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
    const float v = feat_at<SPLIT>(feat, pix, C, c) > 0.f ? s : 0.f;
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

// dker[hw,c,f] = sum_b feat[b,hw,c] * dout[b, c*F+f]; block = 64 channels x 4 batch lanes, fixed-order LDS sum.
template <bool SPLIT>
__global__ __launch_bounds__(256) void sle_bwd_kernel_kernel(const float* __restrict__ dout, int ldo,
                                                             const void* __restrict__ feat, float* __restrict__ dker,
                                                             int B, int HW, int C) {
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

extern "C" int serl_sle_bwd(const float* dout, int ldo, const float* ker, const void* feat, void* dfeat, float* dker,
                            int B, int HW, int C, int F, int feat_split, void* stream) {
  SERL_CHECK_ARG(dout && ker && feat && B >= 1 && HW >= 1 && C >= 1, "sle_bwd: bad args");
  SERL_CHECK_ARG(F == SLE_F, "sle_bwd: num_features must be %d", SLE_F);
  hipStream_t st = (hipStream_t)stream;
  if (dfeat) {
    if (feat_split)
      hipLaunchKernelGGL(sle_bwd_data_kernel<true>, dim3(cdiv(B * C, 256)), dim3(256), 0, st, dout, ldo, ker, feat, dfeat, B, HW, C);
    else
      hipLaunchKernelGGL(sle_bwd_data_kernel<false>, dim3(cdiv(B * C, 256)), dim3(256), 0, st, dout, ldo, ker, feat, dfeat, B, HW, C);
    SERL_CHECK_LAUNCH("sle_bwd_data");
  }
  if (dker) {
    if (feat_split)
      hipLaunchKernelGGL(sle_bwd_kernel_kernel<true>, dim3(cdiv(C, 64), HW), dim3(256), 0, st, dout, ldo, feat, dker, B, HW, C);
    else
      hipLaunchKernelGGL(sle_bwd_kernel_kernel<false>, dim3(cdiv(C, 64), HW), dim3(256), 0, st, dout, ldo, feat, dker, B, HW, C);
    SERL_CHECK_LAUNCH("sle_bwd_kernel");
  }
  return SERL_OK;
}
