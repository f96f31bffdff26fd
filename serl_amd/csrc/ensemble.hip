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
#include "../../include/serl_hip.h"
#include "igemm_bf3.h"

#define ENS_LN_EPS 1e-6f   // flax.linen.LayerNorm default

namespace {

constexpr int EH = 256;                  // hidden width (both layers)
constexpr int EBM = 32;                  // rows per block
constexpr int EBK = 32;                  // K slab
constexpr int ENW = 8, ENT = 64 * ENW;   // waves / threads per block: wave w owns output columns [32 w, 32 w + 32)
constexpr int ENR = EBM / ENW;           // rows per wave in the LayerNorm phases
constexpr int E_LDKW_X = (EBK + 8) / 2;  // dwords per row of the x slab image   [row][k]
constexpr int E_LDKW_H = (EH + 8) / 2;   // dwords per row of the h0 image       [row][k = 256]
constexpr int E_LDW = EH + 4;            // dwords per k-pair row of the weight slab image [k/2][n]
constexpr int E_XPART = EBM * E_LDKW_X, E_HPART = EBM * E_LDKW_H, E_WPART = (EBK / 2) * E_LDW;
constexpr int E_LDC = EH + 4;            // floats per row of the fp32 tile
constexpr int E_LDS = (4 * E_XPART + 2 * E_HPART + 4 * E_WPART + EBM * E_LDC) * 4;   // two x / weight slab buffers

// XCD-aware block -> (member, row block) map of the fused backward kernel.  Every block of a member streams that member's whole
// kernel (256 KB for a hidden layer): workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share one,
// MI355X_MICROARCH.md), so with a plain (row block, member) grid of 8 row blocks every XCD sees every member exactly once
// -- no L2 reuse, E x 256 KB per XCD from the fabric.  Here all row blocks of member e run on XCD slot e % 8: the kernel
// is fetched once per member and the other row blocks hit the XCD's L2.  Grid = 8 * ceil(E / 8) * nrb blocks; blocks
// whose member index falls past E exit at once.  Placement is a speed matter only (no block depends on another).
__device__ __forceinline__ bool ens_block_map(int E, int nrb, int& e, int& row_base) {
  const int b = blockIdx.x, k = b & 7, j = b >> 3;
  const int m = j / nrb, rb = j - m * nrb;
  e = k + 8 * m;
  row_base = rb * EBM;
  return e < E;
}
static inline unsigned ens_grid(int E, int B) { return 8u * (unsigned)cdiv(E, 8) * (unsigned)cdiv(B, EBM); }

__device__ __forceinline__ float ens_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

struct EnsFwdArgs {
  const float* inp; int ld_inp, din, B, E;
  const float *w0, *b0, *g0, *be0, *w1, *b1, *g1, *be1, *hw, *hb;   // member 0 of the launch; member e at e * stride
  long long w0_ms, w1_ms, v_ms, hb_ms;                              // member strides (floats): kernels, 256-vectors, head bias
  int use_ln;
  int width;                                                         // = 256 at run time: the LayerNorm statistics divide by a
                                                                     // run-time N exactly as ln_act_fwd_kernel does (a constant
                                                                     // lets hipcc fuse the division differently: 1-ulp differences)
  float *h0, *h1, *xh0, *xh1, *rs0, *rs1;                           // (E, B, 256) / (E, B); nullptr: not saved
  float* q;                                                          // (E, B)
};

// weight slab [k0, k0 + 32) x 256 of a row-major [K][256] kernel -> registers (k pair kp + 4 i, columns n4 .. n4 + 3).
// Loads are unconditional from a clamped row and stay RAW in the registers; rows past K are zeroed when the slab is
// written to LDS (ens_wstore) -- a select or multiply right behind the load would make the wave wait for it at once, and a
// load behind a runtime condition makes hipcc branch around it and wait for every one separately.
struct WRegs { float4 v[2][2]; };   // 8 waves: k pairs kp + 8 i (i < 2), columns n4 .. n4 + 3
__device__ __forceinline__ void ens_wload(const float* w, int K, int k0, int kp, int n4, WRegs& r) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = k0 + 2 * (kp + 8 * i) + e;
      r.v[i][e] = *reinterpret_cast<const float4*>(w + (long long)min(k, K - 1) * EH + n4);
    }
}
__device__ __forceinline__ void ens_wstore(uint32_t* img, int K, int k0, int kp, int n4, const WRegs& r) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = k0 + 2 * (kp + 8 * i);
    const float m0 = k < K ? 1.f : 0.f, m1 = k + 1 < K ? 1.f : 0.f;
    uint32_t hi[4], lo[4];
    split_pair(r.v[i][0].x * m0, r.v[i][1].x * m1, hi[0], lo[0]);
    split_pair(r.v[i][0].y * m0, r.v[i][1].y * m1, hi[1], lo[1]);
    split_pair(r.v[i][0].z * m0, r.v[i][1].z * m1, hi[2], lo[2]);
    split_pair(r.v[i][0].w * m0, r.v[i][1].w * m1, hi[3], lo[3]);
    uint32_t* d = img + (kp + 8 * i) * E_LDW + n4;
    *reinterpret_cast<uint4*>(d) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *reinterpret_cast<uint4*>(d + E_WPART) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

// one K slab: A fragments from a k-contiguous image (ldkw dwords per row, slab at dword offset a_kdw), B fragments from the
// weight slab image; wave wn owns columns [32 wn, 32 wn + 32)
__device__ __forceinline__ void ens_mma(const uint32_t* ah_img, int a_part, int ldkw, int a_kdw, const uint32_t* w_img, int wn,
                                        int r, int h, f32x16& acc) {
#pragma unroll
  for (int s = 0; s < EBK / 16; ++s) {
    const uint4 qa = *reinterpret_cast<const uint4*>(ah_img + r * ldkw + a_kdw + s * 8 + h * 4);
    const uint4 ql = *reinterpret_cast<const uint4*>(ah_img + a_part + r * ldkw + a_kdw + s * 8 + h * 4);
    const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, qa), al = __builtin_bit_cast(bf16x8_t, ql);
    const uint32_t* d = w_img + (s * 8 + h * 4) * E_LDW + wn * 32 + r;
    const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, make_uint4(d[0], d[E_LDW], d[2 * E_LDW], d[3 * E_LDW]));
    const uint32_t* dl = d + E_WPART;
    const bf16x8_t bl = __builtin_bit_cast(bf16x8_t, make_uint4(dl[0], dl[E_LDW], dl[2 * E_LDW], dl[3 * E_LDW]));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  }
}

__device__ __forceinline__ void ens_acc_to_tile(float* Cs, int wn, int r, int h, const f32x16& acc) {
#pragma unroll
  for (int q = 0; q < 16; ++q) Cs[((q & 3) + 8 * (q >> 2) + 4 * h) * E_LDC + wn * 32 + r] = acc[q];
}

// rows [4 wid, 4 wid + 4) of the fp32 tile: bias, LayerNorm, ReLU exactly as ln_act_fwd_kernel (lane handles columns
// lane + 64 i); y to HBM (hout) and / or to the bf16 hi / lo image of the next layer (himg); head: q = y . hw + hb
__device__ __forceinline__ void ens_ln_rows(const float* Cs, int wid, int lane, int row_base, int B, const float* bias,
                                            const float* gamma, const float* beta, int use_ln, int width, float* hout,
                                            float* xhout,
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
  // the rows of the wave are independent: their reductions are issued together so that the cross-lane steps of one
  // row hide behind those of the others (a rolled row loop is a chain of ~18 dependent shuffles per row)
  float z[ENR][4], s1[ENR], s2[ENR];
#pragma unroll
  for (int rr = 0; rr < ENR; ++rr) {
    const int row = wid * ENR + rr;
    s1[rr] = 0.f; s2[rr] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = 0.f;
      v += Cs[row * E_LDC + lane + 64 * i];
      v += bi[i];
      z[rr][i] = v; s1[rr] += v; s2[rr] = fmaf(v, v, s2[rr]);
    }
  }
  float mean[ENR], mul[ENR];
  if (use_ln) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int rr = 0; rr < ENR; ++rr) {
        s1[rr] += __shfl_xor(s1[rr], o);
        s2[rr] += __shfl_xor(s2[rr], o);
      }
  }
#pragma unroll
  for (int rr = 0; rr < ENR; ++rr) {
    mean[rr] = 0.f; mul[rr] = 1.f;
    if (use_ln) {
      mean[rr] = s1[rr] / (float)width;
      const float var = fmaxf(fmaf(-mean[rr], mean[rr], s2[rr] / (float)width), 0.f);
      mul[rr] = rsqrtf(var + ENS_LN_EPS);
      const int grow = row_base + wid * ENR + rr;
      if (rsout && lane == 0 && grow < B) rsout[grow] = mul[rr];
    }
  }
  float dot[ENR];
#pragma unroll
  for (int rr = 0; rr < ENR; ++rr) {
    const int row = wid * ENR + rr, grow = row_base + row;
    dot[rr] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = lane + 64 * i;
      float v = z[rr][i];
      if (use_ln) {
        const float xh = (v - mean[rr]) * mul[rr];
        if (xhout && grow < B) xhout[(long long)grow * EH + n] = xh;
        v = fmaf(v - mean[rr], mul[rr] * ga[i], be[i]);
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
      dot[rr] = fmaf(v, hv[i], dot[rr]);
    }
  }
  if (hw) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int rr = 0; rr < ENR; ++rr) dot[rr] += __shfl_xor(dot[rr], o);
#pragma unroll
    for (int rr = 0; rr < ENR; ++rr) {
      const int grow = row_base + wid * ENR + rr;
      if (lane == 0 && grow < B) q[grow] = dot[rr] + hb[0];
    }
  }
}

// ---- TanhNormal policy head behind the same two-layer MLP (serl/distributions/tanh_normal.py:17-53; MLP without
// LayerNorm, activate_final): mean = h1 . mk + mb, log_std = clip(h1 . lk + lb, -20, 2), a = tanh(mean + sigma * eps) with
// eps from the jax.random.normal stream at the GLOBAL element index, log pi(a) -- the arithmetic of a GEMM launch for the
// two Dense heads + tanh_gaussian_fwd_kernel, here as the epilogue of the fused forward (POL = true): the policy pass was
// four dependent launches (3 GEMMs + sample), 42 us of the state-based learner's 160 us step.
constexpr int PA = 8;                     // most action dimensions the fused head takes
struct PolHead {
  const float *mk, *lk, *mb, *lb;         // (256, A) kernels, (A) biases
  int A; uint32_t k0, k1; int row_offset, total_rows, deterministic;
  float* act; int ld_act;                 // (B, A) actions, row stride ld_act
  float* ms;                              // (B, 2A) raw [mean | log_std] (what the backward reads) or nullptr
  float* eps; float* logp;                // (B, A) / (B) or nullptr
};
struct PolW { float m[4][PA], l[4][PA]; };
__device__ __forceinline__ void pol_wload(const PolHead& ph, int lane, PolW& w) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < PA; ++j) {        // unconditional loads from a clamped column (columns >= A are never used)
      const int at = (lane + 64 * i) * ph.A + min(j, ph.A - 1);
      w.m[i][j] = ph.mk[at];
      w.l[i][j] = ph.lk[at];
    }
}
// rows [4 wid, 4 wid + 4): h1 = relu(acc + b1) (to HBM when hout), the 2A head dot products, sample + log-prob in lanes j < A
__device__ __forceinline__ void pol_head_rows(const float* Cs, int wid, int lane, int row_base, int B, const float* bias,
                                              float* hout, const PolHead& ph, const PolW& w) {
  float bi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bi[i] = bias[lane + 64 * i];
  float dm[ENR][PA], dl[ENR][PA];
#pragma unroll
  for (int rr = 0; rr < ENR; ++rr) {
    const int row = wid * ENR + rr, grow = row_base + row;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = fmaxf(Cs[row * E_LDC + lane + 64 * i] + bi[i], 0.f);
      if (hout && grow < B) hout[(long long)grow * EH + lane + 64 * i] = v[i];
    }
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      float sm = 0.f, sl = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) { sm = fmaf(v[i], w.m[i][j], sm); sl = fmaf(v[i], w.l[i][j], sl); }
      dm[rr][j] = sm; dl[rr][j] = sl;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int rr = 0; rr < ENR; ++rr)
#pragma unroll
      for (int j = 0; j < PA; ++j) {
        dm[rr][j] += __shfl_xor(dm[rr][j], o);
        dl[rr][j] += __shfl_xor(dl[rr][j], o);
      }
  const int A = ph.A, jl = min(lane, A - 1);
  const float mb = ph.mb[jl], lb = ph.lb[jl];
  float lp[ENR];
#pragma unroll
  for (int rr = 0; rr < ENR; ++rr) {
    const int grow = row_base + wid * ENR + rr;
    float mu = 0.f, raw = 0.f;
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      mu = lane == j ? dm[rr][j] : mu;
      raw = lane == j ? dl[rr][j] : raw;
    }
    mu += mb; raw += lb;
    const bool on = lane < A && grow < B;
    const float ls = fminf(fmaxf(raw, -20.f), 2.f);
    float e = 0.f;
    if (!ph.deterministic) {
      const uint32_t idx = (uint32_t)(ph.row_offset + min(grow, B - 1)) * (uint32_t)A + (uint32_t)jl;
      e = jax_normal_from_bits(threefry_bits_at(ph.k0, ph.k1, idx, (uint32_t)ph.total_rows * (uint32_t)A));
    }
    const float x = mu + expf(ls) * e;
    const float z = (x - mu) * expf(-ls);
    const float term = (-0.5f * z * z - ls - 0.918938533204672742f) - 2.f * (0.693147180559945309f - x - softplusf(-2.f * x));
    lp[rr] = on ? term : 0.f;
    if (on) {
      ph.act[(long long)grow * ph.ld_act + lane] = tanhf(x);
      if (ph.ms) { ph.ms[(long long)grow * 2 * A + lane] = mu; ph.ms[(long long)grow * 2 * A + A + lane] = raw; }
      if (ph.eps) ph.eps[(long long)grow * A + lane] = e;
    }
  }
  if (ph.logp) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1)       // lanes 0 .. A - 1 <= 7 hold the terms
#pragma unroll
      for (int rr = 0; rr < ENR; ++rr) lp[rr] += __shfl_xor(lp[rr], o);
#pragma unroll
    for (int rr = 0; rr < ENR; ++rr) {
      const int grow = row_base + wid * ENR + rr;
      if (lane == 0 && grow < B) ph.logp[grow] = lp[rr];
    }
  }
}

// K slabs of both layers form ONE sequence s = 0 .. n1 + 7 (n1 = ceil(din / 32) slabs of layer 1, then the 8 of layer 2)
// that runs through a ring of four register sets and two LDS buffers: in step s the block multiplies slab s from LDS
// buffer s & 1, writes slab s + 1 (loaded four steps ago) into the other buffer and requests slab s + 5 -- one barrier per
// step, and four steps (~4 us) for a load to land: the blocks are few (E x B / 32) and each walks its K loop alone on its
// CU, so the memory round trip of a step is not hidden by anybody else.  R = n1 % 4 makes every ring index static.
template <int R, bool POL>
__device__ __forceinline__ void ens_fwd_body(const EnsFwdArgs& a, const PolHead& ph, const int e, const int row_base) {
  extern __shared__ __attribute__((aligned(16))) uint32_t e_smem[];
  uint32_t* const Xs = e_smem;                       // [buf][hi, lo][32][E_LDKW_X]
  uint32_t* const Ws = Xs + 4 * E_XPART;             // [buf][hi, lo][16][E_LDW]
  uint32_t* const Hs = Ws + 4 * E_WPART;             // [hi, lo][32][E_LDKW_H]
  float* const Cs = reinterpret_cast<float*>(Hs + 2 * E_HPART);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const float* w0 = a.w0 + e * a.w0_ms;
  const float* w1 = a.w1 + e * a.w1_ms;
  const int kp = tid >> 6, n4 = (tid & 63) * 4;      // weight slab slot: k pairs kp + 8 i, columns n4 .. n4 + 3
  const int xr = tid >> 4, xk = (tid & 15) * 2;      // x slab slot: row xr, k xk, xk + 1
  const int n1 = (a.din + EBK - 1) / EBK;
  const float* xrow = a.inp + (long long)min(row_base + xr, a.B - 1) * a.ld_inp;
  const bool xrow_ok = row_base + xr < a.B;

  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;

  WRegs wr[4];
  float xv[4][2];
  // every load is unconditional (clamped address; zeroed when written to LDS): slabs past the end of the sequence cost a
  // few cached loads
  auto issue = [&](auto set_tag, int sl) {
    constexpr int S_ = decltype(set_tag)::value;
    const bool l1 = sl < n1;
    const int k0 = (l1 ? sl : sl - n1) * EBK;
    ens_wload(l1 ? w0 : w1, l1 ? a.din : EH, k0, kp, n4, wr[S_]);
#pragma unroll
    for (int j = 0; j < 2; ++j) xv[S_][j] = xrow[min(sl * EBK + xk + j, a.din - 1)];
    __builtin_amdgcn_sched_barrier(0);     // keep the loads HERE: hipcc otherwise sinks them towards their use, four steps later
  };
  auto store = [&](auto set_tag, int buf, int sl) {      // slab sl (held by this set) -> LDS buffer buf
    constexpr int S_ = decltype(set_tag)::value;
    const bool l1 = sl < n1;
    ens_wstore(Ws + buf * 2 * E_WPART, l1 ? a.din : EH, (l1 ? sl : sl - n1) * EBK, kp, n4, wr[S_]);
    if (l1) {
      float x[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) x[j] = xv[S_][j] * ((xrow_ok && sl * EBK + xk + j < a.din) ? 1.f : 0.f);
      uint32_t h0, l0;
      split_pair(x[0], x[1], h0, l0);
      uint32_t* d = Xs + buf * 2 * E_XPART + xr * E_LDKW_X + (xk >> 1);
      d[0] = h0;
      d[E_XPART] = l0;
    }
  };
  constexpr std::integral_constant<int, 0> T0{};
  constexpr std::integral_constant<int, 1> T1{};
  constexpr std::integral_constant<int, 2> T2{};
  constexpr std::integral_constant<int, 3> T3{};
  issue(T0, 0); issue(T1, 1); issue(T2, 2); issue(T3, 3);
  store(T0, 0, 0);
  issue(T0, 4);                                       // set 0 is free again: every later step refills the set it stores
  __syncthreads();
  // layer-1 step of slab sl, whose ring position U = sl % 4 is static
  auto step1 = [&](auto u_tag, auto next_tag, int sl) {
    constexpr int U = decltype(u_tag)::value;
    ens_mma(Xs + (U & 1) * 2 * E_XPART, E_XPART, E_LDKW_X, 0, Ws + (U & 1) * 2 * E_WPART, wid, r, h, acc);
    store(next_tag, (U + 1) & 1, sl + 1);
    issue(next_tag, sl + 5);
    __syncthreads();
  };
  int sl = 0;
  for (; sl + 4 <= n1; sl += 4) {
    step1(T0, T1, sl); step1(T1, T2, sl + 1); step1(T2, T3, sl + 2); step1(T3, T0, sl + 3);
  }
  if constexpr (R >= 1) step1(T0, T1, sl);
  if constexpr (R >= 2) step1(T1, T2, sl + 1);
  if constexpr (R >= 3) step1(T2, T3, sl + 2);
  // ---- layer 1 done (and layer 2's first weight slab is already in LDS buffer n1 & 1)
  ens_acc_to_tile(Cs, wid, r, h, acc);
  __syncthreads();
  const long long vo = e * a.v_ms;
  const long long ro = (long long)e * a.B;
  ens_ln_rows(Cs, wid, lane, row_base, a.B, a.b0 + vo, a.use_ln ? a.g0 + vo : nullptr, a.use_ln ? a.be0 + vo : nullptr,
              a.use_ln, a.width, a.h0 ? a.h0 + ro * EH : nullptr, a.xh0 ? a.xh0 + ro * EH : nullptr, a.rs0 ? a.rs0 + ro : nullptr, Hs,
              nullptr, nullptr, nullptr);
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  PolW pw;
  if constexpr (POL) {                                // the head kernels: requested here, used behind layer 2
    pol_wload(ph, lane, pw);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();                                    // the h0 image is complete
  // ---- layer 2: K = 256, A operand resident in LDS; ring position of its slab t is (R + t) % 4
  auto step2 = [&](auto u_tag, auto next_tag, int t) {
    constexpr int U = decltype(u_tag)::value;
    ens_mma(Hs, E_HPART, E_LDKW_H, t * (EBK / 2), Ws + (U & 1) * 2 * E_WPART, wid, r, h, acc);
    if (t + 1 < EH / EBK) store(next_tag, (U + 1) & 1, n1 + t + 1);
    if (t + 5 < EH / EBK) issue(next_tag, n1 + t + 5);
    __syncthreads();
  };
  auto run2 = [&](auto a0, auto a1, auto a2, auto a3) {      // a0 = ring position of layer 2's slab 0
    step2(a0, a1, 0); step2(a1, a2, 1); step2(a2, a3, 2); step2(a3, a0, 3);
    step2(a0, a1, 4); step2(a1, a2, 5); step2(a2, a3, 6); step2(a3, a0, 7);
  };
  if constexpr (R == 0) run2(T0, T1, T2, T3);
  if constexpr (R == 1) run2(T1, T2, T3, T0);
  if constexpr (R == 2) run2(T2, T3, T0, T1);
  if constexpr (R == 3) run2(T3, T0, T1, T2);
  ens_acc_to_tile(Cs, wid, r, h, acc);
  __syncthreads();
  if constexpr (POL) {
    pol_head_rows(Cs, wid, lane, row_base, a.B, a.b1, a.h1, ph, pw);
  } else {
    ens_ln_rows(Cs, wid, lane, row_base, a.B, a.b1 + vo, a.use_ln ? a.g1 + vo : nullptr, a.use_ln ? a.be1 + vo : nullptr,
                a.use_ln, a.width, a.h1 ? a.h1 + ro * EH : nullptr, a.xh1 ? a.xh1 + ro * EH : nullptr, a.rs1 ? a.rs1 + ro : nullptr,
                nullptr, a.hw + vo, a.hb + e * a.hb_ms, a.q + ro);
  }
}


// ================================================================= fused backward
// Backward of the same ensemble from dq (E, B) down to the gradient of the first Dense's output, in ONE launch:
//   g1 = dq . hw[e] * (h1 > 0);  dz1 = LN'(g1; xh1, rs1, gamma1)        (the arithmetic of ln_act_bwd_kernel, rank-1 form)
//   dh0 = dz1 . W1[e]^T                                                 (bf16_3x MFMA, K = 256)
//   g0 = dh0 * (h0 > 0);          dz0 = LN'(g0; xh0, rs0, gamma0)
// replacing ln_act_bwd + GEMM + ln_act_bwd (three dependent launches of a latency-bound chain).  dz1, dz0 (and g1, g0 with
// LayerNorm) go to HBM for the parameter-gradient kernels (dW GEMMs, ens_colsum_kernel below) and the input gradient.
// A block owns 32 rows of one member.  The whole of W1[e] -- 128 registers per thread, 8 contraction slabs x 4 float4 --
// is requested before anything else, so the loads land while the first LayerNorm backward runs; the K loop has no
// global load in it.
struct EnsBwdArgs {
  const float* dq; const float* hw; long long v_ms;     // (E, B); head kernel (member e at e * v_ms, as every 256-vector)
  const float *h1, *xh1, *rs1, *g1s;                    // layer 1: activations (E,B,256), x-hat, 1/sigma (E,B), LN scale
  const float* w1; long long w1_ms;                     // (256, 256) row-major [k][n], member stride
  const float *h0, *xh0, *rs0, *g0s;                    // layer 0
  float *dz1, *g1, *dz0, *g0;                           // outputs (E, B, 256); g1 / g0 only with LayerNorm
  int B, use_ln, width, E;
};

// Inputs of the LayerNorm backward of the wave's rows [4 wid, 4 wid + 4), loaded ahead of their use: global loads retire in
// order, so what a phase needs is requested BEFORE the next burst of loads (the weights) and waited for with a counted
// vmcnt while that burst is still in flight.
struct EnsLnIn { float y[ENR][4], xh[ENR][4], rs[ENR], dq[ENR], ga[4], hv[4]; };
__device__ __forceinline__ void ens_ln_bwd_load(EnsLnIn& in, const float* dq, const float* hw, int wid, int lane, int row_base,
                                                int B, const float* y, const float* xhat, const float* rstd,
                                                const float* gamma, int use_ln) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = lane + 64 * i;
    in.ga[i] = use_ln ? gamma[n] : 1.f;
    in.hv[i] = hw ? hw[n] : 0.f;
  }
#pragma unroll
  for (int rr = 0; rr < ENR; ++rr) {
    const int grow = min(row_base + wid * ENR + rr, B - 1);     // clamped: rows past B are computed and dropped
    in.dq[rr] = dq ? dq[grow] : 0.f;
    in.rs[rr] = use_ln ? rstd[grow] : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = lane + 64 * i;
      in.y[rr][i] = y[(long long)grow * EH + n];
      in.xh[rr][i] = use_ln ? xhat[(long long)grow * EH + n] : 0.f;
    }
  }
}

// rows [4 wid, 4 wid + 4) of the block: g = dy * (y > 0), LayerNorm backward (the arithmetic of ln_act_bwd_kernel); results
// to HBM and (img != nullptr) as the bf16 hi / lo A operand of the next GEMM.  dy = dq[row] * hw[n] (Cs == nullptr) or the
// fp32 tile Cs.
__device__ __forceinline__ void ens_ln_bwd_rows(const EnsLnIn& in, const float* Cs, int wid, int lane, int row_base, int B,
                                                int use_ln, int width, float* dz_out, float* g_out, uint32_t* img) {
  float gv[ENR][4], s1[ENR], s2[ENR];
#pragma unroll
  for (int rr = 0; rr < ENR; ++rr) {
    const int row = wid * ENR + rr;
    s1[rr] = 0.f; s2[rr] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = lane + 64 * i;
      const float d = Cs ? Cs[row * E_LDC + n] : in.dq[rr] * in.hv[i];
      const float g = in.y[rr][i] > 0.f ? d : 0.f;
      if (use_ln) {
        const float dxh = g * in.ga[i];
        s1[rr] += dxh;
        s2[rr] += dxh * in.xh[rr][i];
      }
      gv[rr][i] = g;
    }
  }
  if (use_ln) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int rr = 0; rr < ENR; ++rr) {
        s1[rr] += __shfl_xor(s1[rr], o);
        s2[rr] += __shfl_xor(s2[rr], o);
      }
  }
#pragma unroll
  for (int rr = 0; rr < ENR; ++rr) {
    const int row = wid * ENR + rr, grow = row_base + row;
    const bool ok = grow < B;
    float m1 = 0.f, m2 = 0.f;
    if (use_ln) {
      m1 = s1[rr] / (float)width;
      m2 = s2[rr] / (float)width;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = lane + 64 * i;
      float dz = gv[rr][i];
      if (use_ln) dz = in.rs[rr] * (gv[rr][i] * in.ga[i] - m1 - in.xh[rr][i] * m2);
      if (!ok) dz = 0.f;
      if (ok) {
        dz_out[(long long)grow * EH + n] = dz;
        if (use_ln) g_out[(long long)grow * EH + n] = gv[rr][i];
      }
      if (img) {
        uint32_t hi, lo;
        split_pair(dz, 0.f, hi, lo);
        unsigned short* d = reinterpret_cast<unsigned short*>(img + row * E_LDKW_H) + n;
        d[0] = (unsigned short)hi;
        d[2 * E_HPART] = (unsigned short)lo;
      }
    }
  }
}

constexpr int EB_LDS = (2 * E_HPART + 4 * E_WPART + EBM * E_LDC) * 4;

__global__ __launch_bounds__(ENT, 1) void ens_bwd_kernel(const EnsBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t e_smem[];
  uint32_t* const Hs = e_smem;                        // [hi, lo][32][E_LDKW_H]: dz1 as the A operand
  uint32_t* const Ws = Hs + 2 * E_HPART;              // [buf][hi, lo][16 k pairs][E_LDW]: W1^T slabs
  float* const Cs = reinterpret_cast<float*>(Ws + 4 * E_WPART);
  int e, row_base;
  if (!ens_block_map(a.E, (a.B + EBM - 1) / EBM, e, row_base)) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const long long vo = e * a.v_ms, ro = (long long)e * a.B;
  // ---- layer 1's LayerNorm inputs first (see EnsLnIn), then all of this thread's share of W1[e]: for each of the 8
  // contraction slabs (32 n), rows wr + 64 j, columns 4 c4 .. + 3 of the slab -- 8 consecutive lanes read one 128-byte row
  // segment (full cache lines per wave instruction)
  EnsLnIn in;
  ens_ln_bwd_load(in, a.dq + ro, a.hw + vo, wid, lane, row_base, a.B, a.h1 + ro * EH, a.use_ln ? a.xh1 + ro * EH : nullptr,
                  a.use_ln ? a.rs1 + ro : nullptr, a.use_ln ? a.g1s + vo : nullptr, a.use_ln);
  __builtin_amdgcn_sched_barrier(0);
  const int wr = tid >> 3, c4 = tid & 7;
  const float* wbase = a.w1 + e * a.w1_ms + (long long)wr * EH + 4 * c4;
  float4 wv[8][4];
#pragma unroll
  for (int sl = 0; sl < 8; ++sl)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#ifdef EB_NO_W      // ablation build: no weight traffic
      wv[sl][j] = make_float4(0.01f * (float)c4, 0.f, 0.02f, 0.f);
#else
      wv[sl][j] = *reinterpret_cast<const float4*>(wbase + (long long)(64 * j) * EH + sl * EBK);
#endif
    }
  __builtin_amdgcn_sched_barrier(0);                  // issue the loads here, not at their first use
  // ---- layer 1: dq . hw -> dz1 (HBM + LDS image)
  ens_ln_bwd_rows(in, nullptr, wid, lane, row_base, a.B, a.use_ln, a.width, a.dz1 + ro * EH, a.use_ln ? a.g1 + ro * EH : nullptr, Hs);
  // ---- dh0 = dz1 . W1^T: slab sl = contraction indices n in [32 sl, 32 sl + 32); image [n pair][kk]
  auto store = [&](int sl, int buf) {
    uint32_t* img = Ws + buf * 2 * E_WPART;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t h0, l0, h1, l1;
      split_pair(wv[sl][j].x, wv[sl][j].y, h0, l0);
      split_pair(wv[sl][j].z, wv[sl][j].w, h1, l1);
      uint32_t* d = img + (2 * c4) * E_LDW + wr + 64 * j;
      d[0] = h0; d[E_LDW] = h1;
      d[E_WPART] = l0; d[E_WPART + E_LDW] = l1;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  store(0, 0);
  __syncthreads();                                    // the dz1 image and slab 0 are complete
#pragma unroll
  for (int sl = 0; sl < 8; ++sl) {
    if (sl + 1 < 8) store(sl + 1, (sl + 1) & 1);
    if (sl == 6) {                                    // every weight register is dead: request layer 0's LayerNorm inputs
      ens_ln_bwd_load(in, nullptr, nullptr, wid, lane, row_base, a.B, a.h0 + ro * EH, a.use_ln ? a.xh0 + ro * EH : nullptr,
                      a.use_ln ? a.rs0 + ro : nullptr, a.use_ln ? a.g0s + vo : nullptr, a.use_ln);
      __builtin_amdgcn_sched_barrier(0);
    }
    ens_mma(Hs, E_HPART, E_LDKW_H, sl * (EBK / 2), Ws + (sl & 1) * 2 * E_WPART, wid, r, h, acc);
    __syncthreads();
  }
  ens_acc_to_tile(Cs, wid, r, h, acc);
  __syncthreads();
  // ---- layer 0
  ens_ln_bwd_rows(in, Cs, wid, lane, row_base, a.B, a.use_ln, a.width, a.dz0 + ro * EH, a.use_ln ? a.g0 + ro * EH : nullptr, nullptr);
}

// Batch-column sums of the ensemble's backward in ONE launch (were two launches for the head and one per layer):
//   job 0: d hw[e][n] = sum_b dq[e,b] h1[e,b,n],  d hb[e] = sum_b dq[e,b]
//   job 3 (optional): the first Dense's kernel gradient, see EnsColsumArgs
//   job 1 + l: d LNbias_l = sum_b g_l,  d LNscale_l = sum_b g_l xh_l,  d bias_l = sum_b dz_l      (no LayerNorm: d bias_l only)
// Block (e, job, 64-column quarter): 1024 threads = 16 row groups x 64 columns; the partial sums are added in a fixed order.
struct EnsColsumArgs {
  const float *dq, *h1, *g1, *xh1, *dz1, *g0, *xh0, *dz0;
  float *d_hw, *d_hb, *d_lnb1, *d_lns1, *d_b1, *d_lnb0, *d_lns0, *d_b0;   // member e at e * 256 (d_hb: e)
  int B, use_ln;
  // optional (inp != nullptr, din <= CS_DIN): the first Dense's kernel gradient d_w0[e][m][n] = sum_b inp[b][m] dz0[e,b,n]
  // as din weighted column sums in exact fp32 -- for the state-based learners (din = obs + action = 23) this replaces a
  // GEMM launch of 13 us on the latency-bound chain by ~1 us of FMAs in the layer-0 job
  const float* inp; int ld_inp, din; float* d_w0;
};
constexpr int CS_RG = 16, CS_COLS = 64;     // row groups x columns of a block: 1024 threads
constexpr int CS_DIN = 32;
__global__ __launch_bounds__(CS_RG * CS_COLS) void ens_colsum_kernel(const EnsColsumArgs a) {
  __shared__ float part[3][CS_RG][CS_COLS];
  const int e = blockIdx.x, job = blockIdx.y, cl = threadIdx.x & (CS_COLS - 1), rg = threadIdx.x / CS_COLS;
  const int n = blockIdx.z * CS_COLS + cl;
  const long long ro = (long long)e * a.B;
  if (job == 3) {                              // d_w0[e][m][n] = sum_b inp[b][m] dz0[e,b,n]  (block-uniform branch)
    extern __shared__ float cs_dyn[];
    float* const xs = cs_dyn;                  // [256 rows][CS_DIN]: the input rows of the current chunk
    float* const red = cs_dyn + 256 * CS_DIN;  // [16 row groups][16 m][64 columns]
    float accw[CS_DIN];
#pragma unroll
    for (int m = 0; m < CS_DIN; ++m) accw[m] = 0.f;
    for (int b0 = 0; b0 < a.B; b0 += 256) {
      float z[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {           // this thread's 16 rows of the chunk: loads issued together
        const int b = b0 + rg + CS_RG * u;
        z[u] = a.dz0[(ro + min(b, a.B - 1)) * EH + n];
      }
      __syncthreads();                         // (the previous chunk's xs is no longer read)
      for (int i = threadIdx.x; i < 256 * CS_DIN; i += CS_RG * CS_COLS) {
        const int rr = i / CS_DIN, m = i - rr * CS_DIN, b = b0 + rr;
        xs[i] = (b < a.B && m < a.din) ? a.inp[(long long)b * a.ld_inp + m] : 0.f;   // rows past B weigh 0
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float* x = xs + (rg + CS_RG * u) * CS_DIN;      // wave-uniform row: broadcast reads
#pragma unroll
        for (int m4 = 0; m4 < CS_DIN; m4 += 4) {
          const float4 xv = *reinterpret_cast<const float4*>(x + m4);
          accw[m4] = fmaf(xv.x, z[u], accw[m4]); accw[m4 + 1] = fmaf(xv.y, z[u], accw[m4 + 1]);
          accw[m4 + 2] = fmaf(xv.z, z[u], accw[m4 + 2]); accw[m4 + 3] = fmaf(xv.w, z[u], accw[m4 + 3]);
        }
      }
    }
#pragma unroll
    for (int pass = 0; pass < CS_DIN / 16; ++pass) {
      __syncthreads();
#pragma unroll
      for (int m = 0; m < 16; ++m) red[(rg * 16 + m) * CS_COLS + cl] = accw[pass * 16 + m];
      __syncthreads();
      const int m = pass * 16 + rg;            // this thread finishes (m, column): fixed order over the row groups
      float t = red[(0 * 16 + rg) * CS_COLS + cl];
#pragma unroll
      for (int i = 1; i < CS_RG; ++i) t += red[(i * 16 + rg) * CS_COLS + cl];
      if (m < a.din) a.d_w0[((long long)e * a.din + m) * EH + n] = t;
    }
    return;
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  // rows rg, rg + 16, ...: eight rows per trip with their loads issued together (a rolled loop is one memory round trip
  // per row); rows past B read row B - 1 with weight 0
  const float* p0 = job == 0 ? a.h1 : (job == 1 ? a.g1 : a.g0);
  const float* p1 = job == 0 ? nullptr : (job == 1 ? a.xh1 : a.xh0);
  const float* p2 = job == 0 ? nullptr : (job == 1 ? a.dz1 : a.dz0);
  const bool ln = a.use_ln != 0;
  for (int b0 = rg; b0 < a.B; b0 += 8 * CS_RG) {
    float v0[8], v1[8], v2[8], w[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + u * CS_RG;
      const long long at = (ro + min(b, a.B - 1)) * EH + n;
      w[u] = b < a.B ? 1.f : 0.f;
      v0[u] = (job == 0 || ln) ? p0[at] : 0.f;
      v1[u] = (job != 0 && ln) ? p1[at] : 0.f;
      v2[u] = job != 0 ? p2[at] : a.dq[ro + min(b, a.B - 1)];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (job == 0) {
        const float d = v2[u] * w[u];
        s0 = fmaf(d, v0[u], s0);
        s1 += d;
      } else {
        const float g = v0[u] * w[u];
        s0 += g;
        s1 = fmaf(g, v1[u], s1);
        s2 += v2[u] * w[u];
      }
    }
  }
  part[0][rg][cl] = s0; part[1][rg][cl] = s1; part[2][rg][cl] = s2;
  __syncthreads();
  if (rg != 0) return;
  float t[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    float acc = part[q][0][cl];
#pragma unroll
    for (int i = 1; i < CS_RG; ++i) acc += part[q][i][cl];
    t[q] = acc;
  }
  const long long vo = (long long)e * EH + n;
  if (job == 0) {
    a.d_hw[vo] = t[0];
    if (n == 0) a.d_hb[e] = t[1];
  } else {
    if (ln) {
      (job == 1 ? a.d_lnb1 : a.d_lnb0)[vo] = t[0];
      (job == 1 ? a.d_lns1 : a.d_lns0)[vo] = t[1];
    }
    (job == 1 ? a.d_b1 : a.d_b0)[vo] = t[2];
  }
}

// (plain (row block, member) grid: the ring already spreads a block's weight fetch over its whole run time; the XCD-aware
// map of the backward kernel measured 1 us SLOWER here -- 16 blocks of two members share one XCD's L2 ports)
template <int R, bool POL>
__global__ __launch_bounds__(ENT, 1) void ens_fwd_kernel(const EnsFwdArgs a, const PolHead ph) {
  ens_fwd_body<R, POL>(a, ph, blockIdx.y, blockIdx.x * EBM);
}
// Two independent ensemble forwards of the same input width in ONE launch (the state-based learners' target and critic
// passes, both on the caller's stream: a launch less on their latency-bound chain): members [0, a.E) of `a`, then `b`'s.
template <int R>
__global__ __launch_bounds__(ENT, 1) void ens_fwd2_kernel(const EnsFwdArgs a, const EnsFwdArgs b) {
  const int y = blockIdx.y;
  if (y < a.E) ens_fwd_body<R, false>(a, PolHead{}, y, blockIdx.x * EBM);
  else ens_fwd_body<R, false>(b, PolHead{}, y - a.E, blockIdx.x * EBM);
}

template <bool POL>
static int ens_fwd_launch(const EnsFwdArgs& a, const PolHead& ph, int E, const char* name, void* stream) {
  static_assert(E_LDS <= 160 * 1024, "LDS");
  static bool opted[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), name);
  if (dev >= 0 && dev < 64 && !opted[dev]) {
    const void* ks[4] = {(const void*)ens_fwd_kernel<0, POL>, (const void*)ens_fwd_kernel<1, POL>, (const void*)ens_fwd_kernel<2, POL>,
                         (const void*)ens_fwd_kernel<3, POL>};
    for (const void* k : ks)
      SERL_CHECK_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS), name);
    opted[dev] = true;
  }
  const dim3 grid(cdiv(a.B, EBM), E);
  hipStream_t st = (hipStream_t)stream;
  switch (cdiv(a.din, EBK) % 4) {
    case 0: hipLaunchKernelGGL((ens_fwd_kernel<0, POL>), grid, dim3(ENT), E_LDS, st, a, ph); break;
    case 1: hipLaunchKernelGGL((ens_fwd_kernel<1, POL>), grid, dim3(ENT), E_LDS, st, a, ph); break;
    case 2: hipLaunchKernelGGL((ens_fwd_kernel<2, POL>), grid, dim3(ENT), E_LDS, st, a, ph); break;
    default: hipLaunchKernelGGL((ens_fwd_kernel<3, POL>), grid, dim3(ENT), E_LDS, st, a, ph); break;
  }
  SERL_CHECK_LAUNCH(name);
  return SERL_OK;
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
  EnsFwdArgs a{inp, ld_inp, din, B, E, w0, b0, g0, be0, w1, b1, g1, be1, hw, hb, w0_member_stride, w1_member_stride,
               vec_member_stride, head_bias_member_stride, use_ln, EH, h0, h1, xh0, xh1, rs0, rs1, q};
  return ens_fwd_launch<false>(a, PolHead{}, E, "ensemble_fwd_fused", stream);
}

// TanhNormal(MLP(256, 256)) policy forward in ONE launch (POL epilogue above): z (B, ld_z) whose first din columns are the
// MLP input.  h0 / h1 (B, 256): post-ReLU activations for the backward, or NULL; ms (B, 2A) raw [mean | log_std], eps (B, A),
// logp (B): may be NULL.  key NULL-equivalent: deterministic != 0 -> a = tanh(mean).  "bf16_3x" only, action_dim <= 8.
extern "C" int serl_policy_fwd_fused(const float* z, int ld_z, int din, int B, const float* w0, const float* b0,
                                     const float* w1, const float* b1, const float* mk, const float* lk, const float* mb,
                                     const float* lb, int A, unsigned k0, unsigned k1, int row_offset, int total_rows,
                                     int deterministic, float* act, int ld_act, float* ms, float* eps, float* logp,
                                     float* h0, float* h1, void* stream) {
  SERL_CHECK_ARG(z && w0 && b0 && w1 && b1 && mk && lk && mb && lb && act, "policy_fwd_fused: null pointer");
  SERL_CHECK_ARG(B >= 1 && din >= 1 && ld_z >= din && A >= 1 && A <= PA && ld_act >= A, "policy_fwd_fused: bad shape (action_dim <= %d)", PA);
  SERL_CHECK_ARG(row_offset >= 0 && row_offset + B <= total_rows, "policy_fwd_fused: shard outside the global batch");
  SERL_CHECK_ARG(serl_get_matmul_precision() == 1, "policy_fwd_fused: bf16_3x precision only");
  SERL_CHECK_ARG(((uintptr_t)w0 & 15) == 0 && ((uintptr_t)w1 & 15) == 0, "policy_fwd_fused: kernels must be 16-byte aligned");
  EnsFwdArgs a{z, ld_z, din, B, 1, w0, b0, nullptr, nullptr, w1, b1, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, EH,
               h0, h1, nullptr, nullptr, nullptr, nullptr, nullptr};
  PolHead ph{mk, lk, mb, lb, A, k0, k1, row_offset, total_rows, deterministic, act, ld_act, ms, eps, logp};
  return ens_fwd_launch<true>(a, ph, 1, "policy_fwd_fused", stream);
}

// Backward of serl_ensemble_fwd_fused's network for ALL E members (member stride = one member, the critic's own layout) from
// dq (E, B): dz1 / dz0 = gradients w.r.t. the outputs of the second / first Dense, g1 / g0 = gradients w.r.t. the
// LayerNorm outputs (use_ln only).  Precision "bf16_3x" only.
extern "C" int serl_ensemble_bwd_fused(const float* dq, int B, int E, const float* hw, const float* w1, const float* g0s,
                                       const float* g1s, int use_ln, const float* h0, const float* h1, const float* xh0,
                                       const float* xh1, const float* rs0, const float* rs1, float* dz1, float* g1,
                                       float* dz0, float* g0, void* stream) {
  SERL_CHECK_ARG(dq && hw && w1 && h0 && h1 && dz1 && dz0, "ensemble_bwd_fused: null pointer");
  SERL_CHECK_ARG(B >= 1 && E >= 1, "ensemble_bwd_fused: bad shape");
  SERL_CHECK_ARG(!use_ln || (g0s && g1s && xh0 && xh1 && rs0 && rs1 && g1 && g0), "ensemble_bwd_fused: LayerNorm needs scale, x-hat, 1/sigma and the g outputs");
  SERL_CHECK_ARG(serl_get_matmul_precision() == 1, "ensemble_bwd_fused: bf16_3x precision only");
  SERL_CHECK_ARG(((uintptr_t)w1 & 15) == 0, "ensemble_bwd_fused: kernel must be 16-byte aligned");
  static_assert(EB_LDS <= 160 * 1024, "LDS");
  static bool opted[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), "ensemble_bwd_fused");
  if (dev >= 0 && dev < 64 && !opted[dev]) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)ens_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, EB_LDS), "ensemble_bwd_fused");
    opted[dev] = true;
  }
  EnsBwdArgs a{dq, hw, (long long)EH, h1, xh1, rs1, g1s, w1, (long long)EH * EH, h0, xh0, rs0, g0s, dz1, g1, dz0, g0, B, use_ln, EH, E};
  hipLaunchKernelGGL(ens_bwd_kernel, dim3(ens_grid(E, B)), dim3(ENT), EB_LDS, (hipStream_t)stream, a);
  SERL_CHECK_LAUNCH("ensemble_bwd_fused");
  return SERL_OK;
}

// All batch-column sums of that backward in one launch: gradients of the head (d_hw (E,256), d_hb (E)), of both LayerNorms
// (d_lnb*, d_lns*; use_ln only) and of both Dense biases (d_b*), member e at e * 256.  Fixed summation order.
extern "C" int serl_ensemble_colsum(const float* dq, const float* h1, const float* g1, const float* xh1, const float* dz1,
                                    const float* g0, const float* xh0, const float* dz0, int B, int E, int use_ln,
                                    float* d_hw, float* d_hb, float* d_lnb1, float* d_lns1, float* d_b1, float* d_lnb0,
                                    float* d_lns0, float* d_b0, const float* inp, int ld_inp, int din, float* d_w0,
                                    void* stream) {
  SERL_CHECK_ARG((inp == nullptr) == (d_w0 == nullptr), "ensemble_colsum: give inp and d_w0 together");
  SERL_CHECK_ARG(!inp || (din >= 1 && din <= CS_DIN && ld_inp >= din), "ensemble_colsum: the fused first-kernel gradient takes din <= %d", CS_DIN);
  SERL_CHECK_ARG(dq && h1 && dz1 && dz0 && d_hw && d_hb && d_b1 && d_b0, "ensemble_colsum: null pointer");
  SERL_CHECK_ARG(!use_ln || (g1 && xh1 && g0 && xh0 && d_lnb1 && d_lns1 && d_lnb0 && d_lns0), "ensemble_colsum: LayerNorm operands missing");
  SERL_CHECK_ARG(B >= 1 && E >= 1, "ensemble_colsum: bad shape");
  EnsColsumArgs a{dq, h1, g1, xh1, dz1, g0, xh0, dz0, d_hw, d_hb, d_lnb1, d_lns1, d_b1, d_lnb0, d_lns0, d_b0, B, use_ln,
                  inp, ld_inp, din, d_w0};
  constexpr int dyn = (256 * CS_DIN + CS_RG * 16 * CS_COLS) * 4;
  static bool opted[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), "ensemble_colsum");
  if (dev >= 0 && dev < 64 && !opted[dev]) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)ens_colsum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, dyn), "ensemble_colsum");
    opted[dev] = true;
  }
  hipLaunchKernelGGL(ens_colsum_kernel, dim3(E, inp ? 4 : 3, EH / CS_COLS), dim3(CS_RG * CS_COLS), inp ? dyn : 0, (hipStream_t)stream, a);
  SERL_CHECK_LAUNCH("ensemble_colsum");
  return SERL_OK;
}

// Two ensemble forwards (same batch size, same input width) in one launch: see ens_fwd2_kernel.
static int ens_job_args(const serl_ens_fwd_job* j, EnsFwdArgs& a, const char* name) {
  SERL_CHECK_ARG(j && j->inp && j->w0 && j->b0 && j->w1 && j->b1 && j->hw && j->hb && j->q, "%s: null pointer", name);
  SERL_CHECK_ARG(j->B >= 1 && j->E >= 1 && j->din >= 1 && j->ld_inp >= j->din, "%s: bad shape", name);
  SERL_CHECK_ARG(!j->use_ln || (j->g0 && j->be0 && j->g1 && j->be1), "%s: LayerNorm needs scale and bias", name);
  SERL_CHECK_ARG(((uintptr_t)j->w0 & 15) == 0 && ((uintptr_t)j->w1 & 15) == 0 && j->w0_member_stride % 4 == 0 &&
                 j->w1_member_stride % 4 == 0, "%s: kernels must be 16-byte aligned", name);
  a = EnsFwdArgs{j->inp, j->ld_inp, j->din, j->B, j->E, j->w0, j->b0, j->g0, j->be0, j->w1, j->b1, j->g1, j->be1, j->hw, j->hb,
                 j->w0_member_stride, j->w1_member_stride, j->vec_member_stride, j->head_bias_member_stride, j->use_ln, EH,
                 j->h0, j->h1, j->xh0, j->xh1, j->rs0, j->rs1, j->q};
  return SERL_OK;
}
extern "C" int serl_ensemble_fwd_fused_pair(const serl_ens_fwd_job* ja, const serl_ens_fwd_job* jb, void* stream) {
  EnsFwdArgs a, b;
  if (int e = ens_job_args(ja, a, "ensemble_fwd_fused_pair")) return e;
  if (int e = ens_job_args(jb, b, "ensemble_fwd_fused_pair")) return e;
  SERL_CHECK_ARG(a.B == b.B && a.din == b.din, "ensemble_fwd_fused_pair: the two jobs must share batch size and input width");
  SERL_CHECK_ARG(serl_get_matmul_precision() == 1, "ensemble_fwd_fused_pair: bf16_3x precision only");
  static bool opted[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), "ensemble_fwd_fused_pair");
  if (dev >= 0 && dev < 64 && !opted[dev]) {
    const void* ks[4] = {(const void*)ens_fwd2_kernel<0>, (const void*)ens_fwd2_kernel<1>, (const void*)ens_fwd2_kernel<2>,
                         (const void*)ens_fwd2_kernel<3>};
    for (const void* k : ks)
      SERL_CHECK_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, E_LDS), "ensemble_fwd_fused_pair");
    opted[dev] = true;
  }
  const dim3 grid(cdiv(a.B, EBM), a.E + b.E);
  hipStream_t st = (hipStream_t)stream;
  switch (cdiv(a.din, EBK) % 4) {
    case 0: hipLaunchKernelGGL(ens_fwd2_kernel<0>, grid, dim3(ENT), E_LDS, st, a, b); break;
    case 1: hipLaunchKernelGGL(ens_fwd2_kernel<1>, grid, dim3(ENT), E_LDS, st, a, b); break;
    case 2: hipLaunchKernelGGL(ens_fwd2_kernel<2>, grid, dim3(ENT), E_LDS, st, a, b); break;
    default: hipLaunchKernelGGL(ens_fwd2_kernel<3>, grid, dim3(ENT), E_LDS, st, a, b); break;
  }
  SERL_CHECK_LAUNCH("ensemble_fwd_fused_pair");
  return SERL_OK;
}
