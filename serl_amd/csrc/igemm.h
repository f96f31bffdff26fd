// Generic implicit-GEMM core on the gfx950 exact-fp32 matrix pipe
// (v_mfma_f32_32x32x2_f32: 64 FLOP/clk/SIMD, bit-for-bit an fmaf chain in k order).
//
// C[M,N] = A[M,K] * B[K,N] where A and B are *virtual* matrices described by a Problem
// functor (im2col of an NHWC image, a transposed activation, a weight slice ...).  The
// block stages a BM x BK slab of A and a BK x BN slab of B in LDS as [k][m] / [k][n]
// (rows padded by 4 floats so that k-contiguous writers are at most 2-way conflicted and
// fragment reads -- 32 consecutive floats per half-wave -- are conflict-free), then each
// of the 4 waves walks its TM x TN grid of 32x32 accumulators.
//
// Register-staged double buffering: the global loads of K-step t+1 are issued before the
// MFMAs of step t and written to LDS after them (T14 split of the CDNA guide).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V> struct VecF { float v[V]; };
template <int V> __device__ __forceinline__ void ld_vec(const float* p, VecF<V>& o) {
  if constexpr (V == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    o.v[0] = t.x; o.v[1] = t.y; o.v[2] = t.z; o.v[3] = t.w;
  } else {
    o.v[0] = p[0];
  }
}
template <int V> __device__ __forceinline__ void st_vec(float* p, const VecF<V>& o) {
  if constexpr (V == 4) *reinterpret_cast<float4*>(p) = make_float4(o.v[0], o.v[1], o.v[2], o.v[3]);
  else p[0] = o.v[0];
}
#ifndef IGEMM_MIN_WAVES
#define IGEMM_MIN_WAVES 3
#endif

template <int BM_, int BN_, int BK_, int WGM_, int WGN_>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WGM = WGM_, WGN = WGN_;
  static constexpr int TM = BM_ / WGM_ / 32, TN = BN_ / WGN_ / 32;
  static constexpr int NT = 64 * WGM_ * WGN_;   // threads per block (igemm.h / igemm_bf3.h kernels: 4 waves only)
  static_assert(WGM_ * WGN_ == 4 || WGM_ * WGN_ == 8, "4 or 8 waves per block");
  static_assert(TM * 32 * WGM_ == BM_ && TN * 32 * WGN_ == BN_, "tile/wave mismatch");
  static_assert((BM_ * BK_) % 256 == 0 && (BN_ * BK_) % 256 == 0, "load mapping");
};

// ---- epilogue shared by the fp32 and the bf16x3 kernels.
template <class P, class T>
__device__ __forceinline__ void igemm_epilogue(const P& p, const typename P::Ctx& ctx, f32x16 (&acc)[T::TM][T::TN],
                                               int M, int N, int m0, int n0, int wm, int wn, int r, int h, int split) {
  constexpr int TM = T::TM, TN = T::TN;
  // ---- epilogue: C/D map of the 32x32 tile: col = lane&31, row = (q&3) + 8*(q>>2) + 4*(lane>>5).
  // Column context (bias) is fetched once per column; per-element auxiliary loads (ReLU mask source,
  // accumulate target) are issued for a whole 32x32 tile before the first store so they overlap.
  typename P::OCol ocol[TN];
  int ncol[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    ncol[j] = n0 + wn * (TN * 32) + j * 32 + r;
    p.o_col(ctx, ncol[j] < N ? ncol[j] : 0, ocol[j]);
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    typename P::ORow orow[16];
    float aux[16][TN];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + wm * (TM * 32) + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      p.o_row(ctx, m < M ? m : 0, orow[q]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        aux[q][j] = p.o_aux(ctx, orow[q], (m < M && ncol[j] < N) ? ncol[j] : -1, split);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + wm * (TM * 32) + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
#pragma unroll
      for (int j = 0; j < TN; ++j)
        if (m < M && ncol[j] < N) p.store(ctx, orow[q], ocol[j], ncol[j], split, acc[i][j][q], aux[q][j]);
    }
  }
}

// Problem interface (all __device__):
//   struct Ctx; struct ARow; struct AK; struct BCol; struct BK_;
//   static constexpr bool A_KCONTIG, B_NCONTIG;       (which axis is contiguous in memory)
//   bool init(Ctx&, int z) const;                     (false -> block has nothing to do)
//   int  M(ctx), N(ctx), K(ctx);  int ksplit_len() const;
//   static constexpr int VA, VB (1 or 4): elements per load along the contiguous axis (all-or-nothing valid);
//   void a_row(ctx, m, ARow&); void a_k(ctx, k, kend, AK&);
//   void a_load(ctx, ARow, AK, VecF<VA>& raw, float& mul, VecF<VA>& add);
//   void b_col(ctx, n, BCol&); void b_k(ctx, k, kend, BK_&); void b_load(ctx, BK_, BCol, VecF<VB>& raw, float& mul);
//   a_load / b_load return the RAW value of an unconditional load from a clamped (always valid) address; the
//   element is raw*mul+add (mul = 0 outside the matrix).  The mask is applied when the slab is written to
//   LDS, after the MFMAs, so the prefetch stays in flight behind them -- a select on the loaded value makes
//   hipcc branch around every load with a vmcnt(0) each (CDNA guide, "register or load" trap).
//   epilogue: void o_col(ctx, n, OCol&) (per-column context, e.g. bias; n is valid); void o_row(ctx, m, ORow&);
//   float o_aux(ctx, ORow, n_or_-1, split) (per-element auxiliary LOAD, clamped when n < 0; 0 if unused);
//   void store(ctx, ORow, OCol, n, split, float v, float aux).
template <class P, class T>
__global__ __launch_bounds__(256, IGEMM_MIN_WAVES) void igemm_kernel(const P p) {
  static_assert(T::NT == 256, "igemm_kernel runs 4 waves");
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, TM = T::TM, TN = T::TN;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  __shared__ float As[2 * BK * LDA];
  __shared__ float Bs[2 * BK * LDB];

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

  // ---- per-thread load slots.  VA / VB = elements per load along the memory-contiguous axis (1 or 4:
  // float4 global loads; b128 LDS stores where the LDS row runs along that axis too).
  constexpr int VA = P::VA, VB = P::VB;
  constexpr int A_LANES = (P::A_KCONTIG ? BK : BM) / VA;       // threads along the contiguous axis
  constexpr int A_STEP = 256 / A_LANES;                         // rows (or k's) covered per pass
  constexpr int A_EXT = P::A_KCONTIG ? BM : BK;                 // extent of the strided axis
  constexpr int NA = (A_EXT + A_STEP - 1) / A_STEP;
  constexpr int B_LANES = (P::B_NCONTIG ? BN : BK) / VB;
  constexpr int B_STEP = 256 / B_LANES;
  constexpr int B_EXT = P::B_NCONTIG ? BK : BN;
  constexpr int NB = (B_EXT + B_STEP - 1) / B_STEP;
  typename P::ARow arow[P::A_KCONTIG ? NA : 1];
  typename P::BCol bcol[P::B_NCONTIG ? 1 : NB];
  int a_ml, a_kl, b_nl, b_kl;
  if constexpr (P::A_KCONTIG) {
    a_kl = (tid % A_LANES) * VA;
    a_ml = tid / A_LANES;
#pragma unroll
    for (int i = 0; i < NA; ++i) p.a_row(ctx, m0 + a_ml + i * A_STEP, arow[i]);
  } else {
    a_ml = (tid % A_LANES) * VA;
    a_kl = tid / A_LANES;
    // a wave covers 64 consecutive lanes of ONE k: make that provable so the k-side index math runs on the SALU
    if constexpr (A_LANES % 64 == 0) a_kl = __builtin_amdgcn_readfirstlane(a_kl);
    p.a_row(ctx, m0 + a_ml, arow[0]);
  }
  if constexpr (P::B_NCONTIG) {
    b_nl = (tid % B_LANES) * VB;
    b_kl = tid / B_LANES;
    if constexpr (B_LANES % 64 == 0) b_kl = __builtin_amdgcn_readfirstlane(b_kl);
    p.b_col(ctx, n0 + b_nl, bcol[0]);
  } else {
    b_kl = (tid % B_LANES) * VB;
    b_nl = tid / B_LANES;
#pragma unroll
    for (int i = 0; i < NB; ++i) p.b_col(ctx, n0 + b_nl + i * B_STEP, bcol[i]);
  }
  // when one pass already over-covers the strided axis (small tiles) the surplus threads idle (wave-uniform)
  const bool a_on = (NA * A_STEP == A_EXT) || ((P::A_KCONTIG ? a_ml : a_kl) < A_EXT);
  const bool b_on = (NB * B_STEP == B_EXT) || ((P::B_NCONTIG ? b_kl : b_nl) < B_EXT);

  VecF<VA> areg[NA], aadd[NA];
  VecF<VB> breg[NB];
  float amul[NA], bmul[NB];
  auto gload = [&](int k0) {
    if (a_on) {
      if constexpr (P::A_KCONTIG) {
        typename P::AK ak;
        p.a_k(ctx, k0 + a_kl, kend, ak);
#pragma unroll
        for (int i = 0; i < NA; ++i) p.a_load(ctx, arow[i], ak, areg[i], amul[i], aadd[i]);
      } else {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          typename P::AK ak;
          p.a_k(ctx, k0 + a_kl + i * A_STEP, kend, ak);
          p.a_load(ctx, arow[0], ak, areg[i], amul[i], aadd[i]);
        }
      }
    }
    if (b_on) {
      if constexpr (P::B_NCONTIG) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          typename P::BK_ bk;
          p.b_k(ctx, k0 + b_kl + i * B_STEP, kend, bk);
          p.b_load(ctx, bk, bcol[0], breg[i], bmul[i]);
        }
      } else {
        typename P::BK_ bk;
        p.b_k(ctx, k0 + b_kl, kend, bk);
#pragma unroll
        for (int i = 0; i < NB; ++i) p.b_load(ctx, bk, bcol[i], breg[i], bmul[i]);
      }
    }
  };
  auto lstore = [&](int buf) {
    float* As_ = As + buf * (BK * LDA);
    float* Bs_ = Bs + buf * (BK * LDB);
    if (a_on) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        VecF<VA> v;
#pragma unroll
        for (int j = 0; j < VA; ++j) v.v[j] = fmaf(areg[i].v[j], amul[i], aadd[i].v[j]);
        if constexpr (P::A_KCONTIG) {
#pragma unroll
          for (int j = 0; j < VA; ++j) As_[(a_kl + j) * LDA + a_ml + i * A_STEP] = v.v[j];
        } else {
          st_vec<VA>(&As_[(a_kl + i * A_STEP) * LDA + a_ml], v);
        }
      }
    }
    if (b_on) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        VecF<VB> v;
#pragma unroll
        for (int j = 0; j < VB; ++j) v.v[j] = breg[i].v[j] * bmul[i];
        if constexpr (P::B_NCONTIG) {
          st_vec<VB>(&Bs_[(b_kl + i * B_STEP) * LDB + b_nl], v);
        } else {
#pragma unroll
          for (int j = 0; j < VB; ++j) Bs_[(b_kl + j) * LDB + b_nl + i * B_STEP] = v.v[j];
        }
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // LDS is double-buffered: the slab of step t+1 is written into the other buffer right after the MFMAs of
  // step t, so one barrier per K-step suffices.
  gload(kbeg);
  lstore(0);
  __syncthreads();
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = (k0 + BK) < kend;
    if (more) gload(k0 + BK);
    __builtin_amdgcn_sched_barrier(0);   // loads are issued above, consumed (masked + written to LDS) below the MFMAs
    const float* Ap = As + cur * (BK * LDA) + h * LDA + wm * (TM * 32) + r;
    const float* Bp = Bs + cur * (BK * LDB) + h * LDB + wn * (TN * 32) + r;
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = Ap[(2 * kk) * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bp[(2 * kk) * LDB + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) lstore(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  igemm_epilogue<P, T>(p, ctx, acc, M, N, m0, n0, wm, wn, r, h, split);
}

template <class P, class T>
static inline int igemm_launch(const P& p, int max_m, int max_n, int nsplit, int nz, hipStream_t st, const char* name) {
  const int tiles = cdiv(max_m, T::BM) * cdiv(max_n, T::BN);
  if (tiles <= 0 || nsplit <= 0 || nz <= 0) return SERL_OK;
  dim3 grid(tiles, nsplit, nz);
  hipLaunchKernelGGL((igemm_kernel<P, T>), grid, dim3(256), 0, st, p);
  SERL_CHECK_LAUNCH(name);
  return SERL_OK;
}
