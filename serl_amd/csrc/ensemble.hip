// Fused forward of the critic ensemble  Ensemble(StateActionValue(MLP(hidden 256, 256)))  (serl/networks/ensemble.py:8-22,
// state_action_value.py:11-19, mlp.py:18-35): Dense -> [LayerNorm] -> ReLU -> Dense -> [LayerNorm] -> ReLU -> Dense(1)
// in ONE launch.  The layer-by-layer path (driver.hip ensemble_fwd: GEMM, LayerNorm launch, GEMM, LayerNorm + head launch)
// is four dependent launches of 8-12 us each for 0.3 GFLOP: on the state-based configs and in the "heads" phase of the
// pixel configs the launch chain, not the arithmetic, is the cost.  Here a block owns 32 rows of one member:
//   acc1 = x[32 x din] . W0[e]          K slabs of 32 staged in LDS as bf16 hi / lo (the bf16_3x product of igemm_bf3.h)
//   h0   = relu(LN(acc1 + b0))          through an fp32 tile in LDS, one wave per row (the arithmetic of ln_act_fwd_kernel),
//                                       written to HBM only when the backward needs it, and kept in LDS as the bf16 hi / lo
//                                       A operand of layer 2
//   acc2 = h0 . W1[e];  h1 = relu(LN(acc2 + b1));  q = h1 . hw[e] + hb[e]
// Same products, same k order, same LayerNorm / head reductions as the unfused launches: bit-identical results.
// Precision "bf16_3x" only (the driver keeps the layer-by-layer path at "highest").
#include "igemm_bf3.h"

#define ENS_LN_EPS 1e-6f   // flax.linen.LayerNorm default

namespace {

constexpr int EH = 256;                  // hidden width (both layers)
constexpr int EBM = 32;                  // rows per block
constexpr int EBK = 32;                  // K slab
constexpr int E_LDKW_X = (EBK + 8) / 2;  // dwords per row of the x slab image   [row][k]
constexpr int E_LDKW_H = (EH + 8) / 2;   // dwords per row of the h0 image       [row][k = 256]
constexpr int E_LDW = EH + 4;            // dwords per k-pair row of the weight slab image [k/2][n]
constexpr int E_XPART = EBM * E_LDKW_X, E_HPART = EBM * E_LDKW_H, E_WPART = (EBK / 2) * E_LDW;
constexpr int E_LDC = EH + 4;            // floats per row of the fp32 tile
constexpr int E_LDS = (2 * E_XPART + 2 * E_HPART + 2 * E_WPART + EBM * E_LDC) * 4;

__device__ __forceinline__ float ens_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

struct EnsFwdArgs {
  const float* inp; int ld_inp, din, B;
  const float *w0, *b0, *g0, *be0, *w1, *b1, *g1, *be1, *hw, *hb;   // member 0 of the launch; member e at e * stride
  long long w0_ms, w1_ms, v_ms, hb_ms;                              // member strides (floats): kernels, 256-vectors, head bias
  int use_ln;
  float *h0, *h1, *xh0, *xh1, *rs0, *rs1;                           // (E, B, 256) / (E, B); nullptr: not saved
  float* q;                                                          // (E, B)
};

// weight slab [k0, k0 + 32) x 256 of a row-major [K][256] kernel -> registers (k pair kp + 4 i, columns n4 .. n4 + 3)
struct WRegs { float4 v[4][2]; };
__device__ __forceinline__ void ens_wload(const float* w, int K, int k0, int kp, int n4, WRegs& r) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      // unconditional load from a clamped row, zeroed afterwards: a load behind a runtime condition makes hipcc branch
      // around it and wait for each one separately (one memory round trip per element)
      const int k = k0 + 2 * (kp + 4 * i) + e;
      const float4 v = *reinterpret_cast<const float4*>(w + (long long)min(k, K - 1) * EH + n4);
      const float m = k < K ? 1.f : 0.f;
      r.v[i][e] = make_float4(v.x * m, v.y * m, v.z * m, v.w * m);
    }
}
__device__ __forceinline__ void ens_wstore(uint32_t* img, int kp, int n4, const WRegs& r) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t hi[4], lo[4];
    split_pair(r.v[i][0].x, r.v[i][1].x, hi[0], lo[0]);
    split_pair(r.v[i][0].y, r.v[i][1].y, hi[1], lo[1]);
    split_pair(r.v[i][0].z, r.v[i][1].z, hi[2], lo[2]);
    split_pair(r.v[i][0].w, r.v[i][1].w, hi[3], lo[3]);
    uint32_t* d = img + (kp + 4 * i) * E_LDW + n4;
    *reinterpret_cast<uint4*>(d) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *reinterpret_cast<uint4*>(d + E_WPART) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

// one K slab: A fragments from a k-contiguous image (ldkw dwords per row, slab at dword offset a_k0 / 2), B fragments
// from the weight slab image; wave wn owns columns [64 wn, 64 wn + 64)
__device__ __forceinline__ void ens_mma(const uint32_t* ah_img, int a_part, int ldkw, int a_kdw, const uint32_t* w_img, int wn,
                                        int r, int h, f32x16 (&acc)[2]) {
#pragma unroll
  for (int s = 0; s < EBK / 16; ++s) {
    const uint4 qa = *reinterpret_cast<const uint4*>(ah_img + r * ldkw + a_kdw + s * 8 + h * 4);
    const uint4 ql = *reinterpret_cast<const uint4*>(ah_img + a_part + r * ldkw + a_kdw + s * 8 + h * 4);
    const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, qa), al = __builtin_bit_cast(bf16x8_t, ql);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint32_t* d = w_img + (s * 8 + h * 4) * E_LDW + wn * 64 + j * 32 + r;
      const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, make_uint4(d[0], d[E_LDW], d[2 * E_LDW], d[3 * E_LDW]));
      const uint32_t* dl = d + E_WPART;
      const bf16x8_t bl = __builtin_bit_cast(bf16x8_t, make_uint4(dl[0], dl[E_LDW], dl[2 * E_LDW], dl[3 * E_LDW]));
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
    }
  }
}

__device__ __forceinline__ void ens_acc_to_tile(float* Cs, int wn, int r, int h, const f32x16 (&acc)[2]) {
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) Cs[((q & 3) + 8 * (q >> 2) + 4 * h) * E_LDC + wn * 64 + j * 32 + r] = acc[j][q];
}

// rows [8 wid, 8 wid + 8) of the fp32 tile: bias, LayerNorm, ReLU exactly as ln_act_fwd_kernel (lane handles columns
// lane + 64 i); y to HBM (hout) and / or to the bf16 hi / lo image of the next layer (himg); head: q = y . hw + hb
__device__ __forceinline__ void ens_ln_rows(const float* Cs, int wid, int lane, int row_base, int B, const float* bias,
                                            const float* gamma, const float* beta, int use_ln, float* hout, float* xhout,
                                            float* rsout, uint32_t* himg, const float* hw, const float* hb, float* q) {
  float bi[4], ga[4], be[4], hv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = lane + 64 * i;
    bi[i] = bias[n];
    ga[i] = use_ln ? gamma[n] : 1.f;
    be[i] = use_ln ? beta[n] : 0.f;
    hv[i] = hw ? hw[n] : 0.f;
  }
  // the 8 rows of the wave are independent: their reductions are issued together so that the cross-lane steps of one
  // row hide behind those of the others (a rolled row loop is a chain of ~18 dependent shuffles per row)
  float z[8][4], s1[8], s2[8];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int row = wid * 8 + rr;
    s1[rr] = 0.f; s2[rr] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = 0.f;
      v += Cs[row * E_LDC + lane + 64 * i];
      v += bi[i];
      z[rr][i] = v; s1[rr] += v; s2[rr] += v * v;
    }
  }
  float mean[8], mul[8];
  if (use_ln) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        s1[rr] += __shfl_xor(s1[rr], o);
        s2[rr] += __shfl_xor(s2[rr], o);
      }
  }
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    mean[rr] = 0.f; mul[rr] = 1.f;
    if (use_ln) {
      mean[rr] = s1[rr] / (float)EH;
      const float var = fmaxf(s2[rr] / (float)EH - mean[rr] * mean[rr], 0.f);
      mul[rr] = rsqrtf(var + ENS_LN_EPS);
      const int grow = row_base + wid * 8 + rr;
      if (rsout && lane == 0 && grow < B) rsout[grow] = mul[rr];
    }
  }
  float dot[8];
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int row = wid * 8 + rr, grow = row_base + row;
    dot[rr] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = lane + 64 * i;
      float v = z[rr][i];
      if (use_ln) {
        const float xh = (v - mean[rr]) * mul[rr];
        if (xhout && grow < B) xhout[(long long)grow * EH + n] = xh;
        v = (v - mean[rr]) * (mul[rr] * ga[i]) + be[i];
      }
      v = fmaxf(v, 0.f);
      if (grow >= B) v = 0.f;
      if (hout && grow < B) hout[(long long)grow * EH + n] = v;
      if (himg) {                                  // bf16 hi / lo of the next layer's A operand, element (row, k = n)
        uint32_t hi, lo;
        split_pair(v, 0.f, hi, lo);
        unsigned short* d = reinterpret_cast<unsigned short*>(himg + row * E_LDKW_H) + n;
        d[0] = (unsigned short)hi;
        d[2 * E_HPART] = (unsigned short)lo;
      }
      dot[rr] += v * hv[i];
    }
  }
  if (hw) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) dot[rr] += __shfl_xor(dot[rr], o);
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int grow = row_base + wid * 8 + rr;
      if (lane == 0 && grow < B) q[grow] = dot[rr] + hb[0];
    }
  }
}

__global__ __launch_bounds__(256, 1) void ens_fwd_kernel(const EnsFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t e_smem[];
  uint32_t* const Xs = e_smem;                       // [hi, lo][32][E_LDKW_X]
  uint32_t* const Hs = Xs + 2 * E_XPART;             // [hi, lo][32][E_LDKW_H]
  uint32_t* const Ws = Hs + 2 * E_HPART;             // [hi, lo][16][E_LDW]
  float* const Cs = reinterpret_cast<float*>(Ws + 2 * E_WPART);
  const int e = blockIdx.y, row_base = blockIdx.x * EBM;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const float* w0 = a.w0 + e * a.w0_ms;
  const float* w1 = a.w1 + e * a.w1_ms;
  const int kp = tid >> 6, n4 = (tid & 63) * 4;      // weight slab slot: k pairs kp + 4 i, columns n4 .. n4 + 3
  const int xr = tid >> 3, xk = (tid & 7) * 4;       // x slab slot: row xr, k xk .. xk + 3

  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

  // ---- layer 1: K = din.  Two register sets: the loads of slab t + 2 are issued while slab t is multiplied, so a
  // K-step costs half a memory round trip (the blocks are few and each walks its K loop alone on its CU).
  // The whole 256 x 64-column share of layer 2's kernel (8 slabs, 256 registers of the 512 a wave has here) is requested
  // first and lands while layer 1 runs: layer 2's K loop then has no global load in it.
  WRegs w1r[EH / EBK];
  WRegs wr[2];
  float xv[2][4];
  auto xload = [&](int k0, float (&x)[4]) {
    const int grow = row_base + xr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + xk + j;
      const float v = a.inp[(long long)min(grow, a.B - 1) * a.ld_inp + min(k, a.din - 1)];
      x[j] = (grow < a.B && k < a.din) ? v : 0.f;
    }
  };
  auto xstore = [&](const float (&x)[4]) {
    uint32_t h0, l0, h1, l1;
    split_pair(x[0], x[1], h0, l0);
    split_pair(x[2], x[3], h1, l1);
    uint32_t* d = Xs + xr * E_LDKW_X + (xk >> 1);
    *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(d + E_XPART) = make_uint2(l0, l1);
  };
  xload(0, xv[0]);
  ens_wload(w0, a.din, 0, kp, n4, wr[0]);
  xload(EBK, xv[1]);
  ens_wload(w0, a.din, EBK, kp, n4, wr[1]);
#pragma unroll
  for (int t = 0; t < EH / EBK; ++t) ens_wload(w1, EH, t * EBK, kp, n4, w1r[t]);
  for (int k0 = 0; k0 < a.din; k0 += 2 * EBK) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int kk = k0 + u * EBK;
      if (kk >= a.din) break;                         // block-uniform
      xstore(xv[u]);
      ens_wstore(Ws, kp, n4, wr[u]);
      __syncthreads();
      xload(kk + 2 * EBK, xv[u]);                     // past din: zeros, no memory access
      ens_wload(w0, a.din, kk + 2 * EBK, kp, n4, wr[u]);
      ens_mma(Xs, E_XPART, E_LDKW_X, 0, Ws, wid, r, h, acc);
      __syncthreads();
    }
  }
  ens_acc_to_tile(Cs, wid, r, h, acc);
  __syncthreads();
  const long long vo = e * a.v_ms;
  const long long ro = (long long)e * a.B;
  ens_ln_rows(Cs, wid, lane, row_base, a.B, a.b0 + vo, a.use_ln ? a.g0 + vo : nullptr, a.use_ln ? a.be0 + vo : nullptr,
              a.use_ln, a.h0 ? a.h0 + ro * EH : nullptr, a.xh0 ? a.xh0 + ro * EH : nullptr, a.rs0 ? a.rs0 + ro : nullptr, Hs,
              nullptr, nullptr, nullptr);
  // ---- layer 2: K = 256, A operand resident in LDS
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
#pragma unroll
  for (int t = 0; t < EH / EBK; ++t) {
    ens_wstore(Ws, kp, n4, w1r[t]);
    __syncthreads();                                  // (first pass: also orders the h0 image behind its writers)
    ens_mma(Hs, E_HPART, E_LDKW_H, t * (EBK / 2), Ws, wid, r, h, acc);
    __syncthreads();
  }
  ens_acc_to_tile(Cs, wid, r, h, acc);
  __syncthreads();
  ens_ln_rows(Cs, wid, lane, row_base, a.B, a.b1 + vo, a.use_ln ? a.g1 + vo : nullptr, a.use_ln ? a.be1 + vo : nullptr,
              a.use_ln, a.h1 ? a.h1 + ro * EH : nullptr, a.xh1 ? a.xh1 + ro * EH : nullptr, a.rs1 ? a.rs1 + ro : nullptr,
              nullptr, a.hw + vo, a.hb + e * a.hb_ms, a.q + ro);
}

}  // namespace

// E members (member e at the given strides from the pointers of the first one) on a shared input inp (B, ld_inp) whose
// first din columns are read; hidden width 256 twice; h0 / h1 / xh0 / xh1 / rs0 / rs1 may be NULL (inference passes).
extern "C" int serl_ensemble_fwd_fused(const float* inp, int ld_inp, int din, int B, int E, const float* w0, const float* b0,
                                       const float* g0, const float* be0, const float* w1, const float* b1, const float* g1,
                                       const float* be1, const float* hw, const float* hb, long long w0_member_stride,
                                       long long w1_member_stride, long long vec_member_stride,
                                       long long head_bias_member_stride, int use_ln, float* h0, float* h1, float* xh0,
                                       float* xh1, float* rs0, float* rs1, float* q, void* stream) {
  SERL_CHECK_ARG(inp && w0 && b0 && w1 && b1 && hw && hb && q, "ensemble_fwd_fused: null pointer");
  SERL_CHECK_ARG(B >= 1 && E >= 1 && din >= 1 && ld_inp >= din, "ensemble_fwd_fused: bad shape");
  SERL_CHECK_ARG(!use_ln || (g0 && be0 && g1 && be1), "ensemble_fwd_fused: LayerNorm needs scale and bias");
  SERL_CHECK_ARG(serl_get_matmul_precision() == 1, "ensemble_fwd_fused: bf16_3x precision only");
  SERL_CHECK_ARG(((uintptr_t)w0 & 15) == 0 && ((uintptr_t)w1 & 15) == 0 && w0_member_stride % 4 == 0 && w1_member_stride % 4 == 0,
                 "ensemble_fwd_fused: kernels must be 16-byte aligned");
  static bool opted[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), "ensemble_fwd_fused");
  if (dev >= 0 && dev < 64 && !opted[dev]) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)ens_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS),
                   "ensemble_fwd_fused");
    opted[dev] = true;
  }
  EnsFwdArgs a{inp, ld_inp, din, B, w0, b0, g0, be0, w1, b1, g1, be1, hw, hb, w0_member_stride, w1_member_stride,
               vec_member_stride, head_bias_member_stride, use_ln, h0, h1, xh0, xh1, rs0, rs1, q};
  hipLaunchKernelGGL(ens_fwd_kernel, dim3(cdiv(B, EBM), E), dim3(256), E_LDS, (hipStream_t)stream, a);
  SERL_CHECK_LAUNCH("ensemble_fwd_fused");
  return SERL_OK;
}
