// SpatialLearnedEmbeddings (serl/networks/spatial.py:13-27):
//   out[b, c*F + f] = sum_{hw} feat[b,hw,c] * kernel[hw,c,f]        (F = 8 learned embeddings / channel)
// ~0.1 MMAC per image: VALU kernels with channel-contiguous (coalesced) access.
#include "common.h"

#define SLE_F 8

// thread = (b, c); grid.y splits hw into chunks whose partials are summed in a fixed order by the caller
// (nchunk == 1 for the 7x7 maps of the stride-2 encoder).
__global__ __launch_bounds__(256) void sle_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ ker,
                                                      float* __restrict__ out, int ldo, int B, int HW, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float acc[SLE_F];
#pragma unroll
  for (int f = 0; f < SLE_F; ++f) acc[f] = 0.f;
  const float* fp = feat + (long long)b * HW * C + c;
  const float4* kp = (const float4*)(ker + (long long)c * SLE_F);
  for (int hw = 0; hw < HW; ++hw) {
    const float x = fp[(long long)hw * C];
    const float4 k0 = kp[(long long)hw * C * 2], k1 = kp[(long long)hw * C * 2 + 1];
    acc[0] += x * k0.x; acc[1] += x * k0.y; acc[2] += x * k0.z; acc[3] += x * k0.w;
    acc[4] += x * k1.x; acc[5] += x * k1.y; acc[6] += x * k1.z; acc[7] += x * k1.w;
  }
  float* op = out + (long long)b * ldo + c * SLE_F;
#pragma unroll
  for (int f = 0; f < SLE_F; ++f) op[f] = acc[f];
}

extern "C" int serl_sle_fwd(const float* feat, const float* ker, float* out, int ldo, int B, int HW, int C, int F,
                            void* stream) {
  SERL_CHECK_ARG(feat && ker && out && B >= 1 && HW >= 1 && C >= 1, "sle_fwd: bad args");
  SERL_CHECK_ARG(F == SLE_F, "sle_fwd: num_features must be %d", SLE_F);
  SERL_CHECK_ARG((((uintptr_t)ker) & 15) == 0, "sle_fwd: kernel must be 16-byte aligned");
  hipLaunchKernelGGL(sle_fwd_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, (hipStream_t)stream, feat, ker, out, ldo, B,
                     HW, C);
  SERL_CHECK_LAUNCH("sle_fwd");
  return SERL_OK;
}

// dfeat[b,hw,c] = (feat > 0) * sum_f dout[b, c*F+f] * kernel[hw,c,f]   (ReLU of the last conv folded in)
__global__ __launch_bounds__(256) void sle_bwd_data_kernel(const float* __restrict__ dout, int ldo,
                                                           const float* __restrict__ ker,
                                                           const float* __restrict__ feat, float* __restrict__ dfeat,
                                                           int B, int HW, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float d[SLE_F];
  const float* dp = dout + (long long)b * ldo + c * SLE_F;
#pragma unroll
  for (int f = 0; f < SLE_F; ++f) d[f] = dp[f];
  const long long base = (long long)b * HW * C + c;
  const float4* kp = (const float4*)(ker + (long long)c * SLE_F);
  for (int hw = 0; hw < HW; ++hw) {
    const float4 k0 = kp[(long long)hw * C * 2], k1 = kp[(long long)hw * C * 2 + 1];
    float s = d[0] * k0.x;
    s += d[1] * k0.y; s += d[2] * k0.z; s += d[3] * k0.w;
    s += d[4] * k1.x; s += d[5] * k1.y; s += d[6] * k1.z; s += d[7] * k1.w;
    const long long idx = base + (long long)hw * C;
    dfeat[idx] = feat[idx] > 0.f ? s : 0.f;
  }
}

// dker[hw,c,f] = sum_b feat[b,hw,c] * dout[b, c*F+f]; block = 64 channels x 4 batch lanes, fixed-order LDS sum.
__global__ __launch_bounds__(256) void sle_bwd_kernel_kernel(const float* __restrict__ dout, int ldo,
                                                             const float* __restrict__ feat, float* __restrict__ dker,
                                                             int B, int HW, int C) {
  __shared__ float red[4][64][SLE_F + 1];
  const int cl = threadIdx.x & 63, bl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, hw = blockIdx.y;
  float acc[SLE_F];
#pragma unroll
  for (int f = 0; f < SLE_F; ++f) acc[f] = 0.f;
  if (c < C) {
    for (int b = bl; b < B; b += 4) {
      const float x = feat[((long long)b * HW + hw) * C + c];
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

extern "C" int serl_sle_bwd(const float* dout, int ldo, const float* ker, const float* feat, float* dfeat, float* dker,
                            int B, int HW, int C, int F, void* stream) {
  SERL_CHECK_ARG(dout && ker && feat && B >= 1 && HW >= 1 && C >= 1, "sle_bwd: bad args");
  SERL_CHECK_ARG(F == SLE_F, "sle_bwd: num_features must be %d", SLE_F);
  hipStream_t st = (hipStream_t)stream;
  if (dfeat) {
    hipLaunchKernelGGL(sle_bwd_data_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, st, dout, ldo, ker, feat, dfeat, B,
                       HW, C);
    SERL_CHECK_LAUNCH("sle_bwd_data");
  }
  if (dker) {
    hipLaunchKernelGGL(sle_bwd_kernel_kernel, dim3(cdiv(C, 64), HW), dim3(256), 0, st, dout, ldo, feat, dker, B, HW, C);
    SERL_CHECK_LAUNCH("sle_bwd_kernel");
  }
  return SERL_OK;
}
