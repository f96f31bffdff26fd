// Implicit-GEMM core, "bf16_3x" precision: every fp32 operand x is split into two bf16 terms
// x = hi + lo (hi = bf16_rn(x), lo = bf16_rn(x - hi): 16 mantissa bits kept) and the product is
// accumulated in fp32 as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  on v_mfma_f32_32x32x16_bf16
// (the dropped lo*lo term and the split's rounding are <= 2^-16 relative per product; the same
// algorithm XLA calls BF16_BF16_F32_X3).  Three bf16 MFMAs of 32 cycles replace eight fp32 MFMAs of
// 64 cycles per K=16, so the convolutions move from the fp32-MFMA roof to the HBM roof.
//
// Same Problem interface as igemm.h.  LDS images (hi and lo, double-buffered):
//   k-contiguous source  -> [row][BK+8] bf16      written 4 k at a time (8 B), read as one ds_read_b128
//                           per lane (lane (r,h) takes k = 8h..8h+7 of row r: the MFMA operand map);
//   row-contiguous source-> [BK/2][rows+4] dwords, each dword = (k even, k odd) of one row: a thread loads
//                           two k-rows x 4 rows and writes one b128; a lane reads its 8 k as 4 dwords.
// Both give the lane the same k order, so any mix of the two layouts for A and B is consistent.
#pragma once
#include <type_traits>
#include "igemm.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16_rn(float a, float b) {
  f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// (a, b) -> packed hi pair and packed lo pair (element 0 in the low half)
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf16_rn(a, b);
  const float ha = __uint_as_float(hi << 16), hb = __uint_as_float(hi & 0xffff0000u);
  lo = pack_bf16_rn(a - ha, b - hb);
}

// D = K-steps of register prefetch: the launches of the dense layers are a handful of blocks walking a long K loop, and
// with one step in flight every K-step costs a full memory round trip (~2 us); D = 3 keeps three steps in flight.
template <class P, class T, int D = 1>
__global__ __launch_bounds__(256, 2) void igemm_bf3_kernel(const P p) {
  static_assert(D >= 1 && D <= 3, "prefetch depth");
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, TM = T::TM, TN = T::TN;
  static_assert(BK % 16 == 0, "bf16 MFMA consumes K in steps of 16");
  static_assert(T::NT == 256, "igemm_bf3_kernel runs 4 waves");
  constexpr int VA = P::VA, VB = P::VB;
  constexpr bool AKC = P::A_KCONTIG, BKC = !P::B_NCONTIG;          // operand is k-contiguous in memory
  constexpr int LDKW = (BK + 8) / 2;                                 // dwords per row of a k-contiguous image
  constexpr int A_PART = AKC ? BM * LDKW : (BK / 2) * (BM + 4);     // dwords per (hi | lo) image
  constexpr int B_PART = BKC ? BN * LDKW : (BK / 2) * (BN + 4);
  extern __shared__ __attribute__((aligned(16))) uint32_t bf3_smem[];    // dynamic: may exceed 64 KB
  uint32_t* const As = bf3_smem;                                          // [buf][hi,lo][A_PART]
  uint32_t* const Bs = bf3_smem + 4 * A_PART;                             // [buf][hi,lo][B_PART]

  typename P::Ctx ctx;
  if (!p.init(ctx, blockIdx.z)) return;
  const int M = p.M(ctx), N = p.N(ctx), K = p.K(ctx);
  const int tiles_n = (N + BN - 1) / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  if (m0 >= M) return;
  const int split = blockIdx.y;
  const int kbeg = split * p.ksplit_len();
  const int kend = min(K, kbeg + p.ksplit_len());

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wid / T::WGN, wn = wid - wm * T::WGN;

  // ---- load slots.  k-contiguous operand: threads run along k (V per load), rows strided.
  //      row-contiguous operand: threads run along the rows (V per load), each slot = one k PAIR (2 loads).
  constexpr int A_LANES = (AKC ? BK : BM) / VA;
  constexpr int A_STEP = 256 / A_LANES;
  constexpr int A_EXT = AKC ? BM : BK / 2;
  constexpr int NA = (A_EXT + A_STEP - 1) / A_STEP;
  constexpr int AKP = AKC ? 1 : 2;
  constexpr int B_LANES = (BKC ? BK : BN) / VB;
  constexpr int B_STEP = 256 / B_LANES;
  constexpr int B_EXT = BKC ? BN : BK / 2;
  constexpr int NB = (B_EXT + B_STEP - 1) / B_STEP;
  constexpr int BKP = BKC ? 1 : 2;
  typename P::ARow arow[AKC ? NA : 1];
  typename P::BCol bcol[BKC ? NB : 1];
  int a_ml, a_kl, b_nl, b_kl;   // a_kl / b_kl: k offset (k-contiguous) or k-pair index (row-contiguous)
  if constexpr (AKC) {
    a_kl = (tid % A_LANES) * VA;
    a_ml = tid / A_LANES;
#pragma unroll
    for (int i = 0; i < NA; ++i) p.a_row(ctx, m0 + a_ml + i * A_STEP, arow[i]);
  } else {
    a_ml = (tid % A_LANES) * VA;
    a_kl = tid / A_LANES;
    if constexpr (A_LANES % 64 == 0) a_kl = __builtin_amdgcn_readfirstlane(a_kl);
    p.a_row(ctx, m0 + a_ml, arow[0]);
  }
  if constexpr (BKC) {
    b_kl = (tid % B_LANES) * VB;
    b_nl = tid / B_LANES;
#pragma unroll
    for (int i = 0; i < NB; ++i) p.b_col(ctx, n0 + b_nl + i * B_STEP, bcol[i]);
  } else {
    b_nl = (tid % B_LANES) * VB;
    b_kl = tid / B_LANES;
    if constexpr (B_LANES % 64 == 0) b_kl = __builtin_amdgcn_readfirstlane(b_kl);
    p.b_col(ctx, n0 + b_nl, bcol[0]);
  }
  const bool a_on = (NA * A_STEP == A_EXT) || ((AKC ? a_ml : a_kl) < A_EXT);
  const bool b_on = (NB * B_STEP == B_EXT) || ((BKC ? b_nl : b_kl) < B_EXT);

  VecF<VA> areg_[D][NA][AKP], aadd_[D][NA][AKP];
  VecF<VB> breg_[D][NB][BKP];
  float amul_[D][NA][AKP], bmul_[D][NB][BKP];
  auto gload = [&](auto set_tag, int k0) {
    auto& areg = areg_[decltype(set_tag)::value]; auto& aadd = aadd_[decltype(set_tag)::value];
    auto& breg = breg_[decltype(set_tag)::value];
    auto& amul = amul_[decltype(set_tag)::value]; auto& bmul = bmul_[decltype(set_tag)::value];
    if (a_on) {
      if constexpr (AKC) {
        typename P::AK ak;
        p.a_k(ctx, k0 + a_kl, kend, ak);
#pragma unroll
        for (int i = 0; i < NA; ++i) p.a_load(ctx, arow[i], ak, areg[i][0], amul[i][0], aadd[i][0]);
      } else {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            typename P::AK ak;
            p.a_k(ctx, k0 + 2 * (a_kl + i * A_STEP) + e, kend, ak);
            p.a_load(ctx, arow[0], ak, areg[i][e], amul[i][e], aadd[i][e]);
          }
      }
    }
    if (b_on) {
      if constexpr (BKC) {
        typename P::BK_ bk;
        p.b_k(ctx, k0 + b_kl, kend, bk);
#pragma unroll
        for (int i = 0; i < NB; ++i) p.b_load(ctx, bk, bcol[i], breg[i][0], bmul[i][0]);
      } else {
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            typename P::BK_ bk;
            p.b_k(ctx, k0 + 2 * (b_kl + i * B_STEP) + e, kend, bk);
            p.b_load(ctx, bk, bcol[0], breg[i][e], bmul[i][e]);
          }
      }
    }
  };
  // write V values of one row (k-contiguous) or V rows x one k pair (row-contiguous) as hi/lo bf16
  auto put_kc = [](uint32_t* img, int part, int row, int kl, const float* v, auto vtag) {
    constexpr int V = decltype(vtag)::value;
    if constexpr (V == 4) {
      uint32_t h0, l0, h1, l1;
      split_pair(v[0], v[1], h0, l0);
      split_pair(v[2], v[3], h1, l1);
      uint32_t* d = img + row * LDKW + (kl >> 1);
      *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(d + part) = make_uint2(l0, l1);
    } else {
      uint32_t h0, l0;
      split_pair(v[0], 0.f, h0, l0);
      unsigned short* d = reinterpret_cast<unsigned short*>(img + row * LDKW) + kl;
      d[0] = (unsigned short)h0;
      d[2 * part] = (unsigned short)l0;
    }
  };
  auto put_rc = [](uint32_t* img, int part, int ld, int kp, int row, const float* v0, const float* v1, auto vtag) {
    constexpr int V = decltype(vtag)::value;
    uint32_t hi[V], lo[V];
#pragma unroll
    for (int j = 0; j < V; ++j) split_pair(v0[j], v1[j], hi[j], lo[j]);
    uint32_t* d = img + kp * ld + row;
    if constexpr (V == 4) {
      *reinterpret_cast<uint4*>(d) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
      *reinterpret_cast<uint4*>(d + part) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    } else {
      d[0] = hi[0];
      d[part] = lo[0];
    }
  };
  auto lstore = [&](auto set_tag, int buf) {
    auto& areg = areg_[decltype(set_tag)::value]; auto& aadd = aadd_[decltype(set_tag)::value];
    auto& breg = breg_[decltype(set_tag)::value];
    auto& amul = amul_[decltype(set_tag)::value]; auto& bmul = bmul_[decltype(set_tag)::value];
    uint32_t* As_ = As + buf * (2 * A_PART);
    uint32_t* Bs_ = Bs + buf * (2 * B_PART);
    if (a_on) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        float v[AKP][VA];
#pragma unroll
        for (int e = 0; e < AKP; ++e)
#pragma unroll
          for (int j = 0; j < VA; ++j) v[e][j] = fmaf(areg[i][e].v[j], amul[i][e], aadd[i][e].v[j]);
        if constexpr (AKC) put_kc(As_, A_PART, a_ml + i * A_STEP, a_kl, v[0], std::integral_constant<int, VA>{});
        else put_rc(As_, A_PART, BM + 4, a_kl + i * A_STEP, a_ml, v[0], v[AKP - 1], std::integral_constant<int, VA>{});
      }
    }
    if (b_on) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        float v[BKP][VB];
#pragma unroll
        for (int e = 0; e < BKP; ++e)
#pragma unroll
          for (int j = 0; j < VB; ++j) v[e][j] = breg[i][e].v[j] * bmul[i][e];
        if constexpr (BKC) put_kc(Bs_, B_PART, b_nl + i * B_STEP, b_kl, v[0], std::integral_constant<int, VB>{});
        else put_rc(Bs_, B_PART, BN + 4, b_kl + i * B_STEP, b_nl, v[0], v[BKP - 1], std::integral_constant<int, VB>{});
      }
    }
  };
  // fragment of 32 rows starting at `row0` for MFMA k-step s
  auto frag = [&](const uint32_t* img, bool kc_, int ld, int row0, int s) -> bf16x8_t {
    uint4 q;
    if (kc_) {
      q = *reinterpret_cast<const uint4*>(img + (row0 + r) * LDKW + s * 8 + h * 4);
    } else {
      const uint32_t* d = img + (s * 8 + h * 4) * ld + row0 + r;
      q = make_uint4(d[0], d[ld], d[2 * ld], d[3 * ld]);
    }
    return __builtin_bit_cast(bf16x8_t, q);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  auto mma = [&](int cur) {
    const uint32_t* Ah = As + cur * (2 * A_PART);
    const uint32_t* Bh = Bs + cur * (2 * B_PART);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = frag(Ah, AKC, BM + 4, wm * (TM * 32) + i * 32, s);
        al[i] = frag(Ah + A_PART, AKC, BM + 4, wm * (TM * 32) + i * 32, s);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = frag(Bh, BKC, BN + 4, wn * (TN * 32) + j * 32, s);
        bl[j] = frag(Bh + B_PART, BKC, BN + 4, wn * (TN * 32) + j * 32, s);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };
  constexpr std::integral_constant<int, 0> S0{};
  if constexpr (D == 1) {
    gload(S0, kbeg);
    lstore(S0, 0);
    __syncthreads();
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      const bool more = (k0 + BK) < kend;
      if (more) gload(S0, k0 + BK);
      __builtin_amdgcn_sched_barrier(0);
      mma(cur);
      __builtin_amdgcn_sched_barrier(0);
      if (more) lstore(S0, cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  } else {
    // Register set j holds K-step (t + j) mod D.  Loads are unconditional (a step past kend reads element 0 with a
    // zero multiplier), so the code is straight-line and hipcc counts vmcnt exactly: writing step t+1 to LDS waits
    // for its own loads only, the loads of t+2 .. t+D stay in flight.
    constexpr std::integral_constant<int, 1> S1{};
    constexpr std::integral_constant<int, (D > 2 ? 2 : 0)> S2{};
    gload(S0, kbeg);
    gload(S1, kbeg + BK);
    if constexpr (D > 2) gload(S2, kbeg + 2 * BK);
    lstore(S0, 0);
    __syncthreads();
    int cur = 0, k0 = kbeg;
    auto step = [&](auto jt, auto jn) -> bool {
      gload(jt, k0 + D * BK);                  // set jt was written to LDS one step ago: free
      __builtin_amdgcn_sched_barrier(0);
      mma(cur);
      __builtin_amdgcn_sched_barrier(0);
      lstore(jn, cur ^ 1);
      __syncthreads();
      cur ^= 1;
      k0 += BK;
      return k0 < kend;
    };
    if (k0 < kend) {
      while (true) {
        if (!step(S0, S1)) break;
        if constexpr (D == 2) {
          if (!step(S1, S0)) break;
        } else {
          if (!step(S1, S2)) break;
          if (!step(S2, S0)) break;
        }
      }
    }
  }
  igemm_epilogue<P, T>(p, ctx, acc, M, N, m0, n0, wm, wn, r, h, split);
}

// precision: 0 = exact fp32 MFMA (igemm_kernel), 1 = bf16_3x (igemm_bf3_kernel)
extern "C" int serl_get_matmul_precision();

template <class P, class T>
constexpr int bf3_lds_bytes() {
  constexpr int LDKW = (T::BK + 8) / 2;
  constexpr int A_PART = P::A_KCONTIG ? T::BM * LDKW : (T::BK / 2) * (T::BM + 4);
  constexpr int B_PART = !P::B_NCONTIG ? T::BN * LDKW : (T::BK / 2) * (T::BN + 4);
  return 4 * (A_PART + B_PART) * 4;
}

// T: tile of the fp32 kernel, TB: tile of the bf16_3x kernel (deeper K slab: more bytes in flight per block)
template <class P, class T, class TB = T, int D = 1>
static inline int igemm_launch_prec(const P& p, int max_m, int max_n, int nsplit, int nz, hipStream_t st, const char* name) {
  if (serl_get_matmul_precision() == 0) return igemm_launch<P, T>(p, max_m, max_n, nsplit, nz, st, name);
  const int tiles = cdiv(max_m, TB::BM) * cdiv(max_n, TB::BN);
  if (tiles <= 0 || nsplit <= 0 || nz <= 0) return SERL_OK;
  constexpr int lds = bf3_lds_bytes<P, TB>();
  static_assert(lds <= 160 * 1024, "tile does not fit the 160 KB LDS");
  // one opt-in per kernel and device (device count is small; the attribute call is idempotent)
  static bool opted[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), name);
  if (lds > 48 * 1024 && dev >= 0 && dev < 64 && !opted[dev]) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)igemm_bf3_kernel<P, TB, D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds), name);
    opted[dev] = true;
  }
  dim3 grid(tiles, nsplit, nz);
  hipLaunchKernelGGL((igemm_bf3_kernel<P, TB, D>), grid, dim3(256), lds, st, p);
  SERL_CHECK_LAUNCH(name);
  return SERL_OK;
}
