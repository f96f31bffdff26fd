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

template <int BM_, int BN_, int BK_, int WGM_, int WGN_>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_, WGM = WGM_, WGN = WGN_;
  static constexpr int TM = BM_ / WGM_ / 32, TN = BN_ / WGN_ / 32;
  static_assert(WGM_ * WGN_ == 4, "4 waves per block");
  static_assert(TM * 32 * WGM_ == BM_ && TN * 32 * WGN_ == BN_, "tile/wave mismatch");
  static_assert((BM_ * BK_) % 256 == 0 && (BN_ * BK_) % 256 == 0, "load mapping");
};

// Problem interface (all __device__):
//   struct Ctx; struct ARow; struct AK; struct BCol; struct BK_;
//   static constexpr bool A_KCONTIG, B_NCONTIG;       (which axis is contiguous in memory)
//   bool init(Ctx&, int z) const;                     (false -> block has nothing to do)
//   int  M(ctx), N(ctx), K(ctx);  int ksplit_len() const;
//   void a_row(ctx, m, ARow&); void a_k(ctx, k, kend, AK&); float a_load(ctx, ARow, AK);
//   void b_col(ctx, n, BCol&); void b_k(ctx, k, kend, BK_&); float b_load(ctx, BK_, BCol);
//   void store(ctx, m, n, split, float v);
template <class P, class T>
__global__ __launch_bounds__(256) void igemm_kernel(const P p) {
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, TM = T::TM, TN = T::TN;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int NA = BM * BK / 256, NB = BN * BK / 256;
  __shared__ float As[BK * LDA];
  __shared__ float Bs[BK * LDB];

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

  // ---- per-thread load slots
  typename P::ARow arow[P::A_KCONTIG ? NA : 1];
  typename P::BCol bcol[P::B_NCONTIG ? 1 : NB];
  int a_ml, a_kl, b_nl, b_kl;
  if constexpr (P::A_KCONTIG) {
    a_kl = tid % BK;
    a_ml = tid / BK;
#pragma unroll
    for (int i = 0; i < NA; ++i) p.a_row(ctx, m0 + a_ml + i * (256 / BK), arow[i]);
  } else {
    a_ml = tid % BM;
    a_kl = tid / BM;
    p.a_row(ctx, m0 + a_ml, arow[0]);
  }
  if constexpr (P::B_NCONTIG) {
    b_nl = tid % BN;
    b_kl = tid / BN;
    p.b_col(ctx, n0 + b_nl, bcol[0]);
  } else {
    b_kl = tid % BK;
    b_nl = tid / BK;
#pragma unroll
    for (int i = 0; i < NB; ++i) p.b_col(ctx, n0 + b_nl + i * (256 / BK), bcol[i]);
  }

  float areg[NA], breg[NB];
  auto gload = [&](int k0) {
    if constexpr (P::A_KCONTIG) {
      typename P::AK ak;
      p.a_k(ctx, k0 + a_kl, kend, ak);
#pragma unroll
      for (int i = 0; i < NA; ++i) areg[i] = p.a_load(ctx, arow[i], ak);
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        typename P::AK ak;
        p.a_k(ctx, k0 + a_kl + i * (256 / BM), kend, ak);
        areg[i] = p.a_load(ctx, arow[0], ak);
      }
    }
    if constexpr (P::B_NCONTIG) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        typename P::BK_ bk;
        p.b_k(ctx, k0 + b_kl + i * (256 / BN), kend, bk);
        breg[i] = p.b_load(ctx, bk, bcol[0]);
      }
    } else {
      typename P::BK_ bk;
      p.b_k(ctx, k0 + b_kl, kend, bk);
#pragma unroll
      for (int i = 0; i < NB; ++i) breg[i] = p.b_load(ctx, bk, bcol[i]);
    }
  };
  auto lstore = [&]() {
    if constexpr (P::A_KCONTIG) {
#pragma unroll
      for (int i = 0; i < NA; ++i) As[a_kl * LDA + a_ml + i * (256 / BK)] = areg[i];
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) As[(a_kl + i * (256 / BM)) * LDA + a_ml] = areg[i];
    }
    if constexpr (P::B_NCONTIG) {
#pragma unroll
      for (int i = 0; i < NB; ++i) Bs[(b_kl + i * (256 / BN)) * LDB + b_nl] = breg[i];
    } else {
#pragma unroll
      for (int i = 0; i < NB; ++i) Bs[b_kl * LDB + b_nl + i * (256 / BK)] = breg[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  gload(kbeg);
  lstore();
  __syncthreads();
  const float* Ap = As + h * LDA + wm * (TM * 32) + r;
  const float* Bp = Bs + h * LDB + wn * (TN * 32) + r;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = (k0 + BK) < kend;
    if (more) gload(k0 + BK);
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
    __syncthreads();
    if (more) {
      lstore();
      __syncthreads();
    }
  }

  // ---- epilogue: C/D map of the 32x32 tile: col = lane&31, row = (q&3) + 8*(q>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = m0 + wm * (TM * 32) + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (m < M) {
        typename P::ORow orow;
        p.o_row(ctx, m, orow);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int n = n0 + wn * (TN * 32) + j * 32 + r;
          if (n < N) p.store(ctx, orow, n, split, acc[i][j][q]);
        }
      }
    }
  }
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
