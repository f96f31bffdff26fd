// Implicit GEMM over operands that are ALREADY split into bf16 hi / lo planes ("S format") in HBM.
//
// S format of a [rows][cols] fp32 tensor (cols % 8 == 0): ushort[rows][2][cols] -- for every row the cols
// bf16 "hi" values (hi = bf16_rn(x)) followed by the cols bf16 "lo" values (lo = bf16_rn(x - hi)); the same
// 4 bytes per element as fp32.  The conv stack keeps its activations, activation gradients and (re-laid-out)
// weights in this format at precision "bf16_3x", so the GEMM main loop is pure data movement + MFMA:
//   global 16-B chunk (8 bf16 of one plane) -> registers -> LDS image -> MFMA fragments,
//   acc += a_lo*b_hi + a_hi*b_lo + a_hi*b_hi   (v_mfma_f32_32x32x16_bf16, fp32 accumulate)
// with no per-element conversion, masking or index arithmetic (the in-kernel split of igemm_bf3.h costs
// ~140 VALU instructions per K=16 step and makes that kernel instruction-issue bound).
//
// Operand images in LDS, per plane:
//   KC (chunk = 8 consecutive k of one row)  : [rows][BK+8] bf16, fragment = one ds_read_b128
//   RC (chunk = 8 consecutive rows of one k) : [BK][row stride] bf16, fragment = two ds_read_b64_tr_b16
//       (hardware transpose read: lane i of a 16-lane group gets column i of a 4(k) x 16(rows) block);
//       row stride = 64 B mod 128 B so that the four k-rows of a block fall in disjoint banks.
// Out-of-range chunks are read from a zero page (pointer select, no data select).  uint8 operands (the
// first conv layer's pixels) are loaded as 8 bytes and widened to bf16 after the MFMAs; they have no lo plane.
#pragma once
#include "igemm_bf3.h"

#ifndef SG_DEPTH
#define SG_DEPTH 1   // K-steps of register prefetch (1 or 2); 2 costs 32-64 VGPRs and measured 2% slower overall
#endif
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));   // native vector: stays in registers (HIP's uint4 struct does not)

__device__ uint4 g_zero_page[512];                       // 8 KB of zeros (never written)
__device__ uint4 g_ones_chunk = {0x00003F80u, 0u, 0u, 0u};  // bf16 (1, 0, 0, 0, 0, 0, 0, 0)

struct __attribute__((packed, aligned(1))) U8x8 { unsigned long long v; };

// 4 bytes -> 4 bf16 (exact: integers 0..255)
__device__ __forceinline__ uint2 u8x4_to_bf16(uint32_t v) {
  const uint32_t f0 = __float_as_uint((float)(v & 0xffu)), f1 = __float_as_uint((float)((v >> 8) & 0xffu));
  const uint32_t f2 = __float_as_uint((float)((v >> 16) & 0xffu)), f3 = __float_as_uint((float)(v >> 24));
  return make_uint2(__builtin_amdgcn_perm(f1, f0, 0x07060302u), __builtin_amdgcn_perm(f3, f2, 0x07060302u));
}

constexpr int s_rc_stride(int rows) { return ((2 * rows - 64 + 127) / 128) * 128 + 64; }   // bytes

// Problem interface (all __device__):
//   static constexpr bool A_KC, B_KC (image kind), A_LO, B_LO (operand has a lo plane), A_U8 (A chunks are bytes),
//   XCD_SWZ (XCD-aware block order);
//   Ctx, ARow, AK, BRow, BK_, ORow, OCol;  init / M / N / K / ksplit_len as in igemm.h;
//   KC operand: a_row(ctx, m, ARow&) per row, a_k(ctx, k, kend, AK&) per chunk (k % 8 == 0);
//   RC operand: a_row(ctx, m, ARow&) per 8-row chunk (m % 8 == 0), a_k(ctx, k, kend, AK&) per k;
//   a_ptr(ctx, ARow, AK, const char*& hi, const char*& lo [, int& shift for A_U8]) -- always-valid addresses
//   (zero page when out of range); same for b_*;  epilogue members as in igemm.h, or with S_OUT = true:
//   SCol + s_col(ctx, n, SCol&) (per 8-column context, e.g. bias), s_finish(ctx, ORow, SCol, n, float v[8]) (bias /
//   activation / mask on 8 consecutive columns), s_out(ctx) (ushort* of the S-format output), s_base(ORow) (ushort
//   offset of the row's record: hi plane at base + n, lo plane at base + N + n); MASK_OUT = true adds s_mask(ctx)
//   (byte mask output or nullptr: byte [s_pix(ORow)][n / 8], bit j = element n + j has a non-zero hi plane).
template <class P, class T, int LBUF>
constexpr int sgemm_lds_bytes() {
  constexpr int KC_STRIDE = (T::BK + 8) * 2;
  constexpr int A_PLANE = P::A_KC ? T::BM * KC_STRIDE : T::BK * s_rc_stride(T::BM);
  constexpr int B_PLANE = P::B_KC ? T::BN * KC_STRIDE : T::BK * s_rc_stride(T::BN);
  constexpr int stage = LBUF * ((P::A_LO ? 2 : 1) * A_PLANE + (P::B_LO ? 2 : 1) * B_PLANE);
  constexpr int epi = P::S_OUT ? T::BM * (T::BN + 4) * 4 : 0;       // fp32 tile of the S-format epilogue
  return stage > epi ? stage : epi;
}

// ---- S-format output through LDS (shared by sgemm_kernel and sgemm_ring_kernel): the accumulators (column on the
// lane, rows in registers) are laid out as an fp32 tile in `smem` (BM x (BN + 4) floats; the caller has made sure that
// nobody reads the staging images any more), then every thread finishes 8 consecutive channels of one row: bias / ReLU /
// mask, split into hi and lo, two 16-byte stores -- 8x fewer (and full-width) store instructions than one 2-byte store
// per element.
template <class P, int BM, int BN, int NT, int TM, int TN>
__device__ __forceinline__ void sgemm_s_epilogue(const P& p, const typename P::Ctx& ctx, f32x16 (&acc)[TM][TN],
                                                 unsigned char* smem, int M, int N, int m0, int n0, int wm, int wn,
                                                 int r, int h, int tid) {
  constexpr int LDC = BN + 4;
  float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = wm * (TM * 32) + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        Cs[row * LDC + wn * (TN * 32) + j * 32 + r] = acc[i][j][q];
      }
  __syncthreads();
  constexpr int CH = BN / 8, RPP = NT / CH;          // chunks per row, rows per pass
  const int ch = tid % CH, rr = tid / CH;
  const int n = n0 + ch * 8;
  if (n < N) {
    typename P::SCol sc;
    p.s_col(ctx, n, sc);
    unsigned short* out = p.s_out(ctx);
#pragma unroll 2
    for (int row = rr; row < BM; row += RPP) {
      const int m = m0 + row;
      if (m >= M) break;
      typename P::ORow orow;
      p.o_row(ctx, m, orow);
      const float4 v0 = *reinterpret_cast<const float4*>(Cs + row * LDC + ch * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(Cs + row * LDC + ch * 8 + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      p.s_finish(ctx, orow, sc, n, v);
      u32x4_t hi, lo;
      uint32_t hh, ll;
      split_pair(v[0], v[1], hh, ll); hi[0] = hh; lo[0] = ll;
      split_pair(v[2], v[3], hh, ll); hi[1] = hh; lo[1] = ll;
      split_pair(v[4], v[5], hh, ll); hi[2] = hh; lo[2] = ll;
      split_pair(v[6], v[7], hh, ll); hi[3] = hh; lo[3] = ll;
      unsigned short* dst = out + p.s_base(orow) + n;
      *reinterpret_cast<u32x4_t*>(dst) = hi;
      *reinterpret_cast<u32x4_t*>(dst + N) = lo;
      if constexpr (P::MASK_OUT) {
        // 1 bit per element "hi plane != 0" (the ReLU mask the data gradient of the next layer needs): a byte per
        // 8 channels instead of a 16-byte read of the hi plane (a full 128-byte line per pixel when C = 32)
        if (unsigned char* mk = p.s_mask(ctx)) {
          uint32_t bits = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bits |= (hi[j] & 0x7fffu) ? (1u << (2 * j)) : 0u;
            bits |= (hi[j] & 0x7fff0000u) ? (2u << (2 * j)) : 0u;
          }
          mk[(size_t)p.s_pix(orow) * (N >> 3) + (n >> 3)] = (unsigned char)bits;
        }
      }
    }
  }
}

#ifndef SG_MIN_WAVES
#define SG_MIN_WAVES 2
#endif
template <class P, class T, int LBUF>
__global__ __launch_bounds__(T::NT, SG_MIN_WAVES) void sgemm_kernel(const P p) {
  constexpr int NT = T::NT;
  constexpr int BM = T::BM, BN = T::BN, BK = T::BK, TM = T::TM, TN = T::TN;
  static_assert(BK == 16 || BK == 32 || BK == 64, "K slab");
  static_assert(LBUF == 1 || LBUF == 2, "LDS stages: 2 = one barrier per K-step, 1 = two barriers, half the LDS");
  constexpr bool AKC = P::A_KC, BKC = P::B_KC, ALO = P::A_LO, BLO = P::B_LO, AU8 = P::A_U8;
  constexpr int KC_STRIDE = (BK + 8) * 2;                               // bytes per row of a KC image
  constexpr int A_STRIDE = AKC ? KC_STRIDE : s_rc_stride(BM);
  constexpr int B_STRIDE = BKC ? KC_STRIDE : s_rc_stride(BN);
  constexpr int A_PLANE = AKC ? BM * KC_STRIDE : BK * A_STRIDE;         // bytes
  constexpr int B_PLANE = BKC ? BN * KC_STRIDE : BK * B_STRIDE;
  constexpr int A_BUF = (ALO ? 2 : 1) * A_PLANE, B_BUF = (BLO ? 2 : 1) * B_PLANE;
  extern __shared__ __attribute__((aligned(16))) unsigned char s_smem[];
  unsigned char* const As = s_smem;                     // [buf][plane][...]
  unsigned char* const Bs = s_smem + LBUF * A_BUF;

  // XCD-aware block order: workgroups are dealt round-robin to the 8 XCDs (each with its own 4 MB L2), so blocks
  // that are neighbours in the grid -- the column tiles of one row tile, row tiles sharing an im2col halo, the tiles
  // of one split-K slab -- would never share an L2.  Bijective remap: XCD x gets a contiguous chunk of the grid.
  // Measured: weight gradient -13..16 % (its tiles of one K slab share x and dz), forward neutral, data gradient
  // +25..50 % (parity classes of one region then run at the same time) -- so it is a per-problem switch.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if constexpr (P::XCD_SWZ) {
    const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
    const int lin = bx + gx * (by + gy * bz);
    const int q = nwg >> 3, rem = nwg & 7, xcd = lin & 7;
    const int swz = xcd * q + min(xcd, rem) + (lin >> 3);
    bx = swz % gx;
    const int t = swz / gx;
    by = t % gy; bz = t / gy;
  }
  typename P::Ctx ctx;
  if (!p.init(ctx, bz)) return;
  const int M = p.M(ctx), N = p.N(ctx), K = p.K(ctx);
  const int tiles_n = (N + BN - 1) / BN;
  const int tm = bx / tiles_n, tn = bx - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  if (m0 >= M) return;
  const int split = by;
  const int kbeg = split * p.ksplit_len();
  const int kend = min(K, kbeg + p.ksplit_len());

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wid / T::WGN, wn = wid - wm * T::WGN;

  // ---- chunk slots of this thread
  constexpr int A_DIV = AKC ? BK / 8 : BM / 8;          // chunks along the fast axis of the thread map
  constexpr int A_STEP = NT / A_DIV;
  constexpr int A_EXT = AKC ? BM : BK;
  constexpr int NA = (A_EXT + A_STEP - 1) / A_STEP;
  constexpr int B_DIV = BKC ? BK / 8 : BN / 8;
  constexpr int B_STEP = NT / B_DIV;
  constexpr int B_EXT = BKC ? BN : BK;
  constexpr int NB = (B_EXT + B_STEP - 1) / B_STEP;
  const int a_f = (tid % A_DIV) * 8, a_s = tid / A_DIV;   // fast-axis element offset, slow-axis index
  const int b_f = (tid % B_DIV) * 8, b_s = tid / B_DIV;
  typename P::ARow arow[AKC ? NA : 1];
  typename P::BRow brow[BKC ? NB : 1];
  if constexpr (AKC) {
#pragma unroll
    for (int i = 0; i < NA; ++i) p.a_row(ctx, m0 + a_s + i * A_STEP, arow[i]);
  } else {
    p.a_row(ctx, m0 + a_f, arow[0]);
  }
  if constexpr (BKC) {
#pragma unroll
    for (int i = 0; i < NB; ++i) p.b_row(ctx, n0 + b_s + i * B_STEP, brow[i]);
  } else {
    p.b_row(ctx, n0 + b_f, brow[0]);
  }
  const bool a_on = (NA * A_STEP == A_EXT) || (a_s < A_EXT);
  const bool b_on = (NB * B_STEP == B_EXT) || (b_s < B_EXT);

  // register sets of the prefetch: SG_DEPTH 1 (default) uses set 0 only (loads of step t+1 issued before the MFMAs
  // of step t); SG_DEPTH 2 alternates two sets (loads of t+2 in flight while t is multiplied): measured 2 % slower.
  u32x4_t ahi[2][NA], alo[2][ALO ? NA : 1], bhi[2][NB], blo[2][BLO ? NB : 1];
  unsigned long long a8[2][AU8 ? NA : 1];
  int a8sh[2][AU8 ? NA : 1];
  auto gload = [&](auto set_tag, int k0) {
    constexpr int S_ = decltype(set_tag)::value;
#ifdef SG_NO_LOAD   // ablation build: only the first two K-steps are loaded
    if (k0 >= kbeg + 2 * BK) return;
#endif
    if (a_on) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        typename P::AK ak;
        const char *ph, *pl;
        if constexpr (AKC) {
          p.a_k(ctx, k0 + a_f, kend, ak);
        } else {
          p.a_k(ctx, k0 + a_s + i * A_STEP, kend, ak);
        }
        if constexpr (AU8) {
          p.a_ptr(ctx, arow[AKC ? i : 0], ak, ph, pl, a8sh[S_][i]);
          a8[S_][i] = reinterpret_cast<const U8x8*>(ph)->v;
        } else {
          p.a_ptr(ctx, arow[AKC ? i : 0], ak, ph, pl);
          ahi[S_][i] = *reinterpret_cast<const u32x4_t*>(ph);
          if constexpr (ALO) alo[S_][i] = *reinterpret_cast<const u32x4_t*>(pl);
        }
      }
    }
    if (b_on) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        typename P::BK_ bk;
        const char *ph, *pl;
        if constexpr (BKC) {
          p.b_k(ctx, k0 + b_f, kend, bk);
        } else {
          p.b_k(ctx, k0 + b_s + i * B_STEP, kend, bk);
        }
        p.b_ptr(ctx, brow[BKC ? i : 0], bk, ph, pl);
        bhi[S_][i] = *reinterpret_cast<const u32x4_t*>(ph);
        if constexpr (BLO) blo[S_][i] = *reinterpret_cast<const u32x4_t*>(pl);
      }
    }
  };
  auto lstore = [&](auto set_tag, int buf) {
    constexpr int S_ = decltype(set_tag)::value;
    unsigned char* As_ = As + buf * A_BUF;
    unsigned char* Bs_ = Bs + buf * B_BUF;
    if (a_on) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int off = AKC ? (a_s + i * A_STEP) * KC_STRIDE + a_f * 2 : (a_s + i * A_STEP) * A_STRIDE + a_f * 2;
        if constexpr (AU8) {
          const unsigned long long v = a8[S_][i] >> a8sh[S_][i];
          const uint2 lo2 = u8x4_to_bf16((uint32_t)v), hi2 = u8x4_to_bf16((uint32_t)(v >> 32));
          const u32x4_t w = {lo2.x, lo2.y, hi2.x, hi2.y};
          *reinterpret_cast<u32x4_t*>(As_ + off) = w;
        } else {
          *reinterpret_cast<u32x4_t*>(As_ + off) = ahi[S_][i];
          if constexpr (ALO) *reinterpret_cast<u32x4_t*>(As_ + A_PLANE + off) = alo[S_][i];
        }
      }
    }
    if (b_on) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int off = BKC ? (b_s + i * B_STEP) * KC_STRIDE + b_f * 2 : (b_s + i * B_STEP) * B_STRIDE + b_f * 2;
        *reinterpret_cast<u32x4_t*>(Bs_ + off) = bhi[S_][i];
        if constexpr (BLO) *reinterpret_cast<u32x4_t*>(Bs_ + B_PLANE + off) = blo[S_][i];
      }
    }
  };
  // MFMA fragment (32 rows from row0, k-step s) of one plane
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_cb = (lane & 16);
  auto frag = [&](const unsigned char* img, auto kc_tag, int stride, int row0, int s) -> bf16x8_t {
    if constexpr (decltype(kc_tag)::value) {
      return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(img + (row0 + r) * KC_STRIDE + (s * 2 + h) * 16));
    } else {
      typedef __attribute__((address_space(3))) s16x4_t* lds_p;
      const unsigned char* a0 = img + (s * 16 + 8 * h + tr_q) * stride + (row0 + tr_cb + 4 * tr_p) * 2;
      const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0));
      const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(a0 + 4 * stride));
      typedef short s16x8_t __attribute__((ext_vector_type(8)));
      const s16x8_t v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      return __builtin_bit_cast(bf16x8_t, v);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  auto mma = [&](int cur) {
#ifdef SG_NO_MMA    // ablation build: no LDS reads / MFMAs
    return;
#endif
    const unsigned char* Ah = As + cur * A_BUF;
    const unsigned char* Bh = Bs + cur * B_BUF;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8_t ah[TM], al[ALO ? TM : 1], bh[TN], bl[BLO ? TN : 1];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = frag(Ah, std::integral_constant<bool, AKC>{}, A_STRIDE, wm * (TM * 32) + i * 32, s);
        if constexpr (ALO) al[i] = frag(Ah + A_PLANE, std::integral_constant<bool, AKC>{}, A_STRIDE, wm * (TM * 32) + i * 32, s);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = frag(Bh, std::integral_constant<bool, BKC>{}, B_STRIDE, wn * (TN * 32) + j * 32, s);
        if constexpr (BLO) bl[j] = frag(Bh + B_PLANE, std::integral_constant<bool, BKC>{}, B_STRIDE, wn * (TN * 32) + j * 32, s);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (ALO) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          if constexpr (BLO) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  };
  constexpr std::integral_constant<int, 0> S0{};
  constexpr std::integral_constant<int, 1> S1{};
  if constexpr (LBUF == 1) {
  gload(S0, kbeg);
  lstore(S0, 0);
  __syncthreads();
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    gload(S0, k0 + BK);
    __builtin_amdgcn_sched_barrier(0);
    mma(0);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    lstore(S0, 0);
    __syncthreads();
  }
  } else {
#if SG_DEPTH == 1
  gload(S0, kbeg);
  lstore(S0, 0);
  __syncthreads();
  {
    int cur = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      gload(S0, k0 + BK);
      __builtin_amdgcn_sched_barrier(0);
      mma(cur);
      __builtin_amdgcn_sched_barrier(0);
      lstore(S0, cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
#else
  // Loads and LDS writes are unconditional (a step past kend reads the zero page into a buffer nobody multiplies):
  // with straight-line code hipcc counts vmcnt exactly, so writing set t+1 does not drain the loads of t+2.
  gload(S0, kbeg);
  gload(S1, kbeg + BK);
  lstore(S0, 0);
  __syncthreads();
  int cur = 0;
  for (int k0 = kbeg; k0 < kend; k0 += 2 * BK) {
    // even step: multiply step t; set 1 holds t+1; set 0 is free -> t+2
    gload(S0, k0 + 2 * BK);
    __builtin_amdgcn_sched_barrier(0);
    mma(cur);
    __builtin_amdgcn_sched_barrier(0);
    lstore(S1, cur ^ 1);
    __syncthreads();
    if (k0 + BK >= kend) break;
    // odd step: multiply t+1; set 0 holds t+2; set 1 is free -> t+3
    gload(S1, k0 + 3 * BK);
    __builtin_amdgcn_sched_barrier(0);
    mma(cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    lstore(S0, cur);
    __syncthreads();
  }
#endif
  }
  if constexpr (P::S_OUT) {
    static_assert(BM * (BN + 4) * 4 <= sgemm_lds_bytes<P, T, LBUF>(), "epilogue tile must fit the staging LDS");
    __syncthreads();                                   // everyone is done reading the last K-step's images
    sgemm_s_epilogue<P, BM, BN, NT, TM, TN>(p, ctx, acc, s_smem, M, N, m0, n0, wm, wn, r, h, tid);
  } else {
    igemm_epilogue<P, T>(p, ctx, acc, M, N, m0, n0, wm, wn, r, h, split);
  }
}

template <class P, class T, int LBUF = 2>
static inline int sgemm_launch(const P& p, int max_m, int max_n, int nsplit, int nz, hipStream_t st, const char* name) {
  const int tiles = cdiv(max_m, T::BM) * cdiv(max_n, T::BN);
  if (tiles <= 0 || nsplit <= 0 || nz <= 0) return SERL_OK;
  constexpr int lds = sgemm_lds_bytes<P, T, LBUF>();
  static_assert(lds <= 160 * 1024, "tile does not fit the 160 KB LDS");
  static bool opted[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), name);
  if (lds > 48 * 1024 && dev >= 0 && dev < 64 && !opted[dev]) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)sgemm_kernel<P, T, LBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds), name);
    opted[dev] = true;
  }
  dim3 grid(tiles, nsplit, nz);
  hipLaunchKernelGGL((sgemm_kernel<P, T, LBUF>), grid, dim3(T::NT), lds, st, p);
  SERL_CHECK_LAUNCH(name);
  return SERL_OK;
}
