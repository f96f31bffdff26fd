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
struct SleBwdArgs { const float* dout[SLE_MAXG]; const float* ker[SLE_MAXG]; const void* feat[SLE_MAXG]; void* dfeat[SLE_MAXG]; float* dker[SLE_MAXG]; };

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
  for (int i = 0; i < ngroups; ++i) {
    SERL_CHECK_ARG(dout[i] && ker[i] && feat[i] && (!dfeat || dfeat[i]) && (!dker || dker[i]), "sle_bwd: null pointer");
    a.dout[i] = dout[i]; a.ker[i] = ker[i]; a.feat[i] = feat[i];
    a.dfeat[i] = dfeat ? dfeat[i] : nullptr; a.dker[i] = dker ? dker[i] : nullptr;
  }
  if (dfeat) {
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
