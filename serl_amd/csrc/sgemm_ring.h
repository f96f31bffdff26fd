// Implicit GEMM on S-format operands (sgemm.h) with an LDS-DMA ring instead of register staging.
//
// sgemm_kernel moves every operand chunk global -> registers -> LDS and synchronises twice per K-step; on the deep
// layers of the conv stack (K = 576 / 1152, N = 128 / 256) its load path and its MFMA path barely overlap (DESIGN.md 8,
// ablation).  Here the operands go global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write)
// into a ring of NSTAGE K-slabs of 32; the DMA of slab t + NSTAGE - 1 is issued right behind the barrier that opens
// slab t and stays in flight across the next barriers (counted s_waitcnt vmcnt, raw s_barrier), so a block needs ONE
// barrier per K-step and HBM/L2 latency is hidden inside the block instead of by a second resident block.
//
// LDS image of a stage: [A hi][A lo][B hi][B lo], each plane [rows][32 k] bf16 = 64 bytes per row, unpadded -- the
// destination of a DMA wave-instruction is 64 consecutive 16-byte slots (16 rows x 4 chunks), so the swizzle that makes
// the fragment reads conflict-free is applied to the per-lane SOURCE address: slot (row, j) holds the row's logical
// chunk j ^ ((row >> 2) & 3).  A ds_read_b128 fragment read (lane (r, h) reads logical chunk 2 s + h of row r) is served
// in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: the rows of a group have four different
// (row >> 2) & 3 for each row & 3, so its 16 slots are 16 different 16-byte columns of the 256-byte bank row.
// Same k order and MFMA sequence per accumulator as sgemm_kernel: the outputs are bit-identical to it.
#pragma once
#include "sgemm.h"

template <int BM_, int BN_, int WGM_, int WGN_, int NSTAGE_>
struct RingCfg {
  static constexpr int BM = BM_, BN = BN_, BK = 32, WGM = WGM_, WGN = WGN_, NSTAGE = NSTAGE_;
  static constexpr int NW = WGM_ * WGN_, NT = 64 * NW;
  static constexpr int TM = BM_ / WGM_ / 32, TN = BN_ / WGN_ / 32;
  static constexpr int ROWB = BK * 2;                         // bytes per row and plane
  static constexpr int A_PLANE = BM_ * ROWB, B_PLANE = BN_ * ROWB;
  static constexpr int STAGE = 2 * (A_PLANE + B_PLANE);
  static constexpr int PA = BM_ / 16, PB = BN_ / 16;          // 1-KiB DMA pieces (16 rows) per plane
  static constexpr int NAR = PA / NW, NBR = PB / NW;          // row slots per lane
  static constexpr int PPW = 2 * (NAR + NBR);                 // DMA instructions per wave and K-step
  static constexpr int EPI = BM_ * (BN_ + 4) * 4;             // fp32 tile of the S-format epilogue
  static constexpr int LDS = NSTAGE_ * STAGE > EPI ? NSTAGE_ * STAGE : EPI;
  static_assert(TM * 32 * WGM_ == BM_ && TN * 32 * WGN_ == BN_, "tile / wave mismatch");
  static_assert(PA % NW == 0 && PB % NW == 0, "every wave issues the same number of DMA pieces per plane");
  static_assert(NSTAGE_ == 2 || NSTAGE_ == 3, "ring depth");
  static_assert(LDS <= 160 * 1024, "does not fit the 160 KB LDS");
};

typedef __attribute__((address_space(1))) const void* ring_gptr_t;
typedef __attribute__((address_space(3))) void* ring_lptr_t;

template <int N> __device__ __forceinline__ void ring_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <class P, class T>
__global__ __launch_bounds__(T::NT, 1) void sgemm_ring_kernel(const P p) {
  static_assert(P::A_KC && P::B_KC && P::A_LO && P::B_LO && !P::A_U8 && P::S_OUT, "KC x KC operands with lo planes, S-format output");
  constexpr int NW = T::NW, TM = T::TM, TN = T::TN, BK = T::BK, NSTAGE = T::NSTAGE, D = NSTAGE - 1;
  constexpr int A_PLANE = T::A_PLANE, B_PLANE = T::B_PLANE, STAGE = T::STAGE, ROWB = T::ROWB;
  extern __shared__ __attribute__((aligned(16))) unsigned char r_smem[];

  typename P::Ctx ctx;
  if (!p.init(ctx, blockIdx.z)) return;
  const int M = p.M(ctx), N = p.N(ctx), K = p.K(ctx);
  const int tiles_n = (N + T::BN - 1) / T::BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * T::BM, n0 = tn * T::BN;
  if (m0 >= M) return;

  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wid / T::WGN, wn = wid - wm * T::WGN;

  // ---- DMA slots of this lane: row lane/4 of the wave's 16-row pieces, logical chunk (lane & 3) ^ swizzle(row)
  const int lrow = lane >> 2, lch = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
  typename P::ARow arow[T::NAR];
  typename P::BRow brow[T::NBR];
#pragma unroll
  for (int i = 0; i < T::NAR; ++i) p.a_row(ctx, m0 + (i * NW + wid) * 16 + lrow, arow[i]);
#pragma unroll
  for (int i = 0; i < T::NBR; ++i) p.b_row(ctx, n0 + (i * NW + wid) * 16 + lrow, brow[i]);

  auto issue = [&](int stage, int k0) {
    unsigned char* const sb = r_smem + stage * STAGE + wid * 1024;
    typename P::AK ak;
    typename P::BK_ bk;
    p.a_k(ctx, k0 + lch, K, ak);
    p.b_k(ctx, k0 + lch, K, bk);
#pragma unroll
    for (int i = 0; i < T::NAR; ++i) {
      const char *ph, *pl;
      p.a_ptr(ctx, arow[i], ak, ph, pl);
#if defined(RING_ZERO_SRC) || defined(RING_ZERO_A)   // ablation build: every DMA reads the zero page (L1/L2 hits)
      ph = reinterpret_cast<const char*>(g_zero_page) + (lane & 7) * 16; pl = ph + 128;
#endif
      __builtin_amdgcn_global_load_lds((ring_gptr_t)ph, (ring_lptr_t)(sb + i * NW * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((ring_gptr_t)pl, (ring_lptr_t)(sb + A_PLANE + i * NW * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < T::NBR; ++i) {
      const char *ph, *pl;
      p.b_ptr(ctx, brow[i], bk, ph, pl);
#if defined(RING_ZERO_SRC) || defined(RING_ZERO_B)
      ph = reinterpret_cast<const char*>(g_zero_page) + (lane & 7) * 16; pl = ph + 128;
#endif
      __builtin_amdgcn_global_load_lds((ring_gptr_t)ph, (ring_lptr_t)(sb + 2 * A_PLANE + i * NW * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((ring_gptr_t)pl, (ring_lptr_t)(sb + 2 * A_PLANE + B_PLANE + i * NW * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  // fragment offset of this lane inside a 32-row tile, k16 step 0 (step 1: ^ 32)
  const int fro = r * ROWB + ((h ^ ((r >> 2) & 3)) << 4);
  const int a_off = wm * (TM * 32) * ROWB + fro, b_off = 2 * A_PLANE + wn * (TN * 32) * ROWB + fro;
  auto mma = [&](int stage) {
    const unsigned char* const sb = r_smem + stage * STAGE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8_t ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const unsigned char* q = sb + ((a_off + i * 32 * ROWB) ^ (s << 5));
        ah[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(q));
        al[i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(q + A_PLANE));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned char* q = sb + ((b_off + j * 32 * ROWB) ^ (s << 5));
        bh[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(q));
        bl[j] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(q + B_PLANE));
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

  const int nk = (K + BK - 1) / BK;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nk) issue(d, d * BK);
  int cur = 0, nxt = D % NSTAGE;
  for (int t = 0; t < nk; ++t) {
    // slab t has landed: this wave's DMAs of it by the counted wait (the slabs issued later stay in flight), everyone
    // else's by the barrier -- which also says that every wave is done reading slab t - 1, whose stage is refilled next
    if (D == 2 && t + 1 < nk) ring_wait_vm<T::PPW>();
    else ring_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
#ifdef RING_NO_ISSUE   // ablation build: no DMA inside the loop
    if (false)
#endif
    if (t + D < nk) issue(nxt, (t + D) * BK);
#ifdef RING_NO_MMA     // ablation build: no LDS reads / MFMAs
    if (t == 0)
#endif
    mma(cur);
    cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
  }
  __syncthreads();                                     // no DMA in flight here; everyone is done with the last slab
#ifdef RING_NO_EPI     // ablation build: no output
  if (acc[0][0][0] != 12345.f) return;
#endif
  sgemm_s_epilogue<P, T::BM, T::BN, T::NT, TM, TN>(p, ctx, acc, r_smem, M, N, m0, n0, wm, wn, r, h, tid);
}

template <class P, class T>
static inline int sgemm_ring_launch(const P& p, int max_m, int max_n, int nz, hipStream_t st, const char* name) {
  const int tiles = cdiv(max_m, T::BM) * cdiv(max_n, T::BN);
  if (tiles <= 0 || nz <= 0) return SERL_OK;
  static bool opted[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), name);
  if (dev >= 0 && dev < 64 && !opted[dev]) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)sgemm_ring_kernel<P, T>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS), name);
    opted[dev] = true;
  }
  hipLaunchKernelGGL((sgemm_ring_kernel<P, T>), dim3(tiles, 1, nz), dim3(T::NT), T::LDS, st, p);
  SERL_CHECK_LAUNCH(name);
  return SERL_OK;
}
