// "Patch" convolutions for the 3x3 stride-2 layers of the conv encoder on S-format tensors (sgemm.h).
//
// The implicit-GEMM kernels of convs.hip gather every tap of the im2col matrix from global memory: each input line is
// fetched 2.25 times (9 taps / stride^2), and because the blocks in flight on an XCD touch ~20 MB against a 4 MB L2
// most of those re-reads miss (measured 1.5-1.7x the algorithmic bytes on the L2-miss side, profiles/r01_v23_traffic.json).
// Here a block owns a band of R output rows of ONE image: it copies the 2R+1 input rows it needs into LDS once (a
// contiguous span of global memory, since every row is taken at full width), and all nine taps of the K loop read their
// MFMA A fragments from that patch.  Input over-fetch is (2R+1)/2R by construction.  Same math, same k order and the
// same MFMA sequence per k16 step as sgemm_kernel<ConvFwdS>: the outputs are bit-identical to it.
//
// Patch layout in LDS: record = one pixel = hi[C] lo[C] bf16 (4C bytes, as in global memory), records of a row split by
// the parity of ix ([row][ix & 1][ix >> 1]) so that the 32 (or 16) lanes of a fragment read -- consecutive ox, i.e. ix
// two apart -- touch consecutive records; the 16-byte chunk index inside a record is XOR-swizzled with the record index
// so that 16 consecutive records land in 16 distinct 16-byte bank groups (ds_read_b128 serves 16 lanes per pass).
// One persistent block of 8 waves per CU walks a contiguous run of (image, band) items of one group with TWO patch
// buffers: the LDS-DMA (global_load_lds_dwordx4, no staging registers) of item i+1 is in flight while item i is
// multiplied and stored, so HBM latency is hidden inside the block.
// Weights (B operand): every wave
// keeps the MFMA B fragments of its N tile for the WHOLE K range in registers (9C/16 * 8 VGPRs): the weight matrix is
// read once per block with coalesced loads, transposed into fragment order through the (still empty) LDS, and the K loop
// has no global load and no barrier in it.  (Streaming a k16 weight slab through LDS per step exposed an L2 round trip
// per step, loading fragments straight from the [F][2][K] layout made every lane fetch its own cache line: both measured
// slower than the im2col kernel.)
#pragma once
#include <stdlib.h>
#include "conv_geom.h"
#include "sgemm.h"

struct PConvFwdArgs {
  const unsigned short* x[MAXG];
  const unsigned short* wf[MAXG];     // S format [F][2][9C], k = (ky, kx, c)
  const float* bias[MAXG];
  unsigned short* y[MAXG];
  unsigned char* rmask[MAXG];
  int B, IH, IW, OH, OW, nbands, relu;
  int wfrag = 0;                      // the fragment-ordered image (pfw_wq_index) follows wf's 2 * F * Kp values
  FastDiv d_iw, d_bands;
};
// position (bf16 units) of weight (k, f) of a plane in pconv_fwdw_kernel's fragment-ordered image:
// [wave = f / 32][k16 step][plane][lane = 32 (k % 16 / 8) + f % 32][k % 8] -- a wave reads its 2 KB per step as one span
__host__ __device__ inline long long pfw_wq_index(int k, int f, int plane, int Kp) {
  return ((((long long)(f >> 5) * (Kp >> 4) + (k >> 4)) * 2 + plane) * 64 + ((k >> 3) & 1) * 32 + (f & 31)) * 8 + (k & 7);
}

// C, F: channels in / out; R: output rows per item; OWP: padded output width (power of two, 32 % OWP == 0).
// 8 waves = (M tiles) x (N tiles) of an item, one 32 x 32 accumulator per wave.
template <int C_, int F_, int R_, int OWP_>
struct PFwdCfg {
  static constexpr int C = C_, F = F_, R = R_, OWP = OWP_;
  static constexpr int NT = 512;
  static constexpr int REC = 4 * C;                 // bytes per pixel record
  static constexpr int NCH = C / 4;                 // 16-byte chunks per record
  static constexpr int NR = 2 * R + 1;              // patch rows
  static constexpr int ROWB = 2 * OWP * REC;        // bytes per patch row (two parity regions of OWP records)
  static constexpr int PATCH = NR * ROWB;
  static constexpr int SLOTS = PATCH / 16;
  static constexpr int MT = (R * OWP + 31) / 32;    // 32-row MFMA tiles
  static constexpr int NTL = F / 32;
  static constexpr int KP = 9 * C;
  static constexpr int WROW = KP * 2 + 16;          // bytes per row of the weight transposition image [plane][f][K]: odd in 16-B units
  static constexpr int WIMG = 2 * F * WROW;
  static constexpr int EPI = MT * 32 * (F + 4) * 4;
  static constexpr int LDS = 2 * PATCH;
  static_assert(MT * NTL == 8, "one accumulator tile per wave, 8 waves");
  static_assert(C % 16 == 0 && F % 32 == 0 && 32 % OWP == 0 && SLOTS % NT == 0, "shape");
  static_assert((KP / 8) % 2 == 0 && ((WROW / 16) & 1) == 1, "weight image rows");
  static_assert(EPI <= PATCH && WIMG <= LDS && LDS <= 160 * 1024, "LDS");
};

// CUs a persistent grid spreads over (all of them; SERL_PCONV_CUS caps it for experiments with co-running streams)
static thread_local int g_pconv_cu_limit = 0;   // serl_convs_set_cu_limit: the calling THREAD's cap for the launches that follow (0: none)
static int pconv_cus(int ncu) {
  static int cap = -1;
  if (cap < 0) { const char* e = getenv("SERL_PCONV_CUS"); cap = e ? atoi(e) : 0; }
  int n = ncu > 0 ? ncu : 256;
  if (g_pconv_cu_limit > 0 && g_pconv_cu_limit < n) n = g_pconv_cu_limit;
  return cap > 0 && cap < n ? cap : n;
}

typedef __attribute__((address_space(1))) const void* pconv_gptr_t;
typedef __attribute__((address_space(3))) void* pconv_lptr_t;

template <int REC> __device__ __forceinline__ int pconv_swz(int j) { return REC == 128 ? ((j >> 1) & 7) : (j & 15); }

// write-only dump area for lanes that must issue a store although their pixel does not exist (pconv_fwdw_kernel: the waits count stores)
__device__ uint4 g_store_dump[192];

// workgroup barrier that waits for this wave's LDS operations only: an LDS-DMA in flight (vmcnt) stays in flight
__device__ __forceinline__ void pconv_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// LDS reads by hand (pconv_fwdw_kernel, pconv_fused12pp_kernel): with an LDS-DMA in flight hipcc's wait-count pass treats lgkmcnt as unordered
// (it models global_load_lds as a flat access) and turns every wait for an LDS read into lgkmcnt(0), i.e. it also waits
// for the prefetches issued since.  These are issued and NOT waited for; the counted s_waitcnt that follows names the
// registers as read-write operands, so no use can move above it.  Offsets are immediates.
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
#define PF_LD_D(d01, d2, addr, imm)                                                                                     \
  asm volatile("ds_read2_b32 %0, %2 offset0:%3 offset1:%4\n\tds_read_b32 %1, %2 offset:%5"                               \
               : "=&v"(d01), "=&v"(d2) : "v"(addr), "n"((imm) / 4), "n"((imm) / 4 + 1), "n"((imm) + 8) : "memory")
#define PF_LD_2X128(o0, o1, a0, a1, imm0, imm1)                                                                         \
  asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%5"                                          \
               : "=&v"(o0), "=&v"(o1) : "v"(a0), "v"(a1), "n"(imm0), "n"(imm1) : "memory")

template <class G>
__global__ __launch_bounds__(512, 1) void pconv_fwd_kernel(const PConvFwdArgs a) {
  constexpr int C = G::C, F = G::F, R = G::R, OWP = G::OWP, NT = G::NT;
  constexpr int REC = G::REC, NCH = G::NCH, ROWB = G::ROWB, PARB = OWP * REC;
  constexpr int RPT = 32 / OWP;
  constexpr int KS = C / 16, NS = 9 * KS;
  constexpr int Kp = G::KP, WROW = G::WROW;
  extern __shared__ __attribute__((aligned(16))) unsigned char p_smem[];

  const int z = blockIdx.z;
  const int nitems = a.B * a.nbands;
  const int per = (nitems + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it0 = blockIdx.x * per, it1 = min(nitems, it0 + per);
  if (it0 >= it1) return;                                   // block-uniform, before any barrier
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wid / G::NTL, wn = wid - wm * G::NTL;

  // ---- weights: contiguous [F][2][Kp] bf16 -> LDS image [plane][f][Kp (+8)] -> B fragments of this wave's N tile:
  // lane (r, h) holds k = 16 s + 8 h .. + 7 of column f = 32 wn + r, for every k16 step s
  bf16x8_t rbh[NS], rbl[NS];
  {
    constexpr int CPR = Kp / 8;                             // 16-byte chunks per (f, plane) row
    constexpr int NWC = 2 * F * CPR, NWL = (NWC + NT - 1) / NT;
    const char* wsrc = reinterpret_cast<const char*>(a.wf[z]);
    u32x4_t wv[NWL];
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int q = tid + i * NT;
      wv[i] = *reinterpret_cast<const u32x4_t*>(q < NWC ? wsrc + (size_t)q * 16 : reinterpret_cast<const char*>(g_zero_page));
    }
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int q = tid + i * NT;
      if (q < NWC) {
        const int row = q / CPR, kc = q - row * CPR;        // row = f * 2 + plane
        *reinterpret_cast<u32x4_t*>(p_smem + ((row & 1) * F + (row >> 1)) * WROW + kc * 16) = wv[i];
      }
    }
    __syncthreads();
    const unsigned char* wrow = p_smem + (wn * 32 + r) * WROW + h * 16;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      rbh[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(wrow + s * 32));
      rbl[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(wrow + F * WROW + s * 32));
    }
    __syncthreads();
  }

  // ---- fragment addresses of this lane (the same for every item)
  const int ox = r & (OWP - 1);
  const int arow = 2 * min(wm * RPT + r / OWP, R - 1) * ROWB;     // padding rows repeat the last row (never stored)
  const int rec0 = ox * REC, rec1 = (ox + 1) * REC;
  const int sw0 = pconv_swz<REC>(ox) << 4, sw1 = pconv_swz<REC>(ox + 1) << 4;
  constexpr int LDC = F + 4;
  constexpr int CH = F / 8, RPP = NT / CH;
  const int n = (tid % CH) * 8;
  float bs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bs[j] = a.bias[z][n + j];
  unsigned short* const out = a.y[z];
  unsigned char* const mk = a.rmask[z];

  // LDS-DMA of one item's patch: rows 2*oy0 .. 2*oy0 + 2*rows of image b are a contiguous span of global memory.  The
  // destination of a wave-instruction is 64 consecutive 16-byte slots, so the parity split and the chunk swizzle are
  // applied to the per-lane SOURCE address: slot l = (row, parity, j, swizzled chunk) in the patch's linear order.
  constexpr int NDMA = G::SLOTS / NT;
  struct DmaSrc { const char* src; int nrows; };
  auto dma_src = [&](int item) {
    const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
    const int oy0 = band * R;
    DmaSrc d;
    d.nrows = 2 * min(R, a.OH - oy0) + 1;
    d.src = reinterpret_cast<const char*>(a.x[z]) + ((size_t)(b * a.IH + 2 * oy0) * a.IW) * REC;
    return d;
  };
  auto dma_piece = [&](const DmaSrc& d, unsigned char* patch, int i) {      // piece i of NDMA: 8 waves x 64 slots
    const int l = (i * 8 + wid) * 64 + lane;
    const int rec = l / NCH, chp = l & (NCH - 1);
    const int rr = rec / (2 * OWP), par = (rec / OWP) & 1, j = rec & (OWP - 1);
    const int ix = 2 * j + par;
    const bool ok = rr < d.nrows && ix < a.IW;
    const char* g = ok ? d.src + (size_t)(((rr * a.IW + ix) * NCH + (chp ^ pconv_swz<REC>(j))) * 16)
                       : reinterpret_cast<const char*>(g_zero_page);
    __builtin_amdgcn_global_load_lds((pconv_gptr_t)g, (pconv_lptr_t)(patch + (i * 8 + wid) * 1024), 16, 0, 0);
  };

  {
    const DmaSrc d0 = dma_src(it0);
#pragma unroll 1
    for (int i = 0; i < NDMA; ++i) dma_piece(d0, p_smem, i);
  }
  int pb = 0;
  for (int item = it0; item < it1; ++item, pb ^= 1) {
    unsigned char* const patch = p_smem + pb * G::PATCH;
    const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
    const int oy0 = band * R;
    const int rows_here = min(R, a.OH - oy0);
    // this item's patch has landed (vmcnt(0) of every wave -- written out: hipcc's __syncthreads() leaves the wait to
    // whichever LDS read comes first, and a wave whose lanes all skip the output loop has none), and every wave is
    // done with the previous item's output tile, which lives in the other buffer
    // The stores of the previous item are younger than this patch's DMA and vmcnt retires in order, so the wait leaves
    // exactly those stores in flight (2, or 3 with a byte mask, per pass).
    static_assert(G::MT * 32 / RPP == 2, "store count of the counted wait");
    if (item == it0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (mk != nullptr) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the next item's DMA pieces are issued between the taps of the K loop: their address arithmetic (VALU) and issue
    // slots hide under the MFMAs instead of forming a phase of their own
    const bool more = item + 1 < it1;
    const DmaSrc dn = dma_src(more ? item + 1 : item);
    unsigned char* const npatch = p_smem + (pb ^ 1) * G::PATCH;

    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#ifdef PCONV_NO_MMA
    if (item == it0)
#endif
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int tapoff = arow + ky * ROWB + (kx & 1) * PARB + (kx == 2 ? rec1 : rec0);
      const int sw = kx == 2 ? sw1 : sw0;
#pragma unroll
      for (int kc = 0; kc < KS; ++kc) {
        const int chh = ((kc * 2 + h) << 4) ^ sw, chl = ((kc * 2 + h + C / 8) << 4) ^ sw;
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(patch + tapoff + chh));
        const bf16x8_t al = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(patch + tapoff + chl));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, rbh[tap * KS + kc], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, rbl[tap * KS + kc], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, rbh[tap * KS + kc], acc, 0, 0, 0);
      }
#ifndef PCONV_NO_DMA
      if (more) {
#pragma unroll
        for (int i = tap * NDMA / 9; i < (tap + 1) * NDMA / 9; ++i) dma_piece(dn, npatch, i);
      }
#endif
    }
    pconv_lds_barrier();                   // every wave is done with the patch before the output tile overwrites it

    // ---- S-format output through an fp32 tile in LDS (as sgemm.h): 8 consecutive channels of one pixel per thread
    float* const Cs = reinterpret_cast<float*>(patch);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      Cs[row * LDC + wn * 32 + r] = acc[q];
    }
    pconv_lds_barrier();
    // Every thread runs every pass and issues the same number of stores: rows past the band's end and the padding
    // column are redirected to the last valid row / column (they store that pixel's values a second time), so the
    // wait at the top of the next item can be a COUNT that leaves exactly these stores in flight.
    const uint32_t cs_base = (uint32_t)(size_t)(pconv_lptr_t)Cs;
#pragma unroll
    for (int pass = 0; pass < G::MT * 32 / RPP; ++pass) {
      const int row0 = tid / CH + pass * RPP;
      const int oyl = min(row0 / OWP, rows_here - 1), oxx = min(row0 & (OWP - 1), a.OW - 1);
      const int row = oyl * OWP + oxx;
      const size_t pix = (size_t)(b * a.OH + oy0 + oyl) * a.OW + oxx;
      // LDS reads by hand: hipcc puts vmcnt(0) in front of a read it cannot tell apart from the DMA's destination,
      // which would wait for the NEXT item's patch here
      u32x4_t c0, c1;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(c0), "=&v"(c1) : "v"(cs_base + (uint32_t)(row * LDC + n) * 4u) : "memory");
      float v[8] = {__uint_as_float(c0[0]), __uint_as_float(c0[1]), __uint_as_float(c0[2]), __uint_as_float(c0[3]),
                    __uint_as_float(c1[0]), __uint_as_float(c1[1]), __uint_as_float(c1[2]), __uint_as_float(c1[3])};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = fmaf(v[j], 1.0f, bs[j]);
        if (a.relu) v[j] = fmaxf(v[j], 0.f);
      }
      u32x4_t hi, lo;
      uint32_t hh, ll;
      split_pair(v[0], v[1], hh, ll); hi[0] = hh; lo[0] = ll;
      split_pair(v[2], v[3], hh, ll); hi[1] = hh; lo[1] = ll;
      split_pair(v[4], v[5], hh, ll); hi[2] = hh; lo[2] = ll;
      split_pair(v[6], v[7], hh, ll); hi[3] = hh; lo[3] = ll;
      unsigned short* dst = out + pix * 2 * F + n;
#ifdef PCONV_NO_STORE
      if (item != it0) continue;
#endif
      *reinterpret_cast<u32x4_t*>(dst) = hi;
      *reinterpret_cast<u32x4_t*>(dst + F) = lo;
      if (mk != nullptr) {
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bits |= (hi[j] & 0x7fffu) ? (1u << (2 * j)) : 0u;
          bits |= (hi[j] & 0x7fff0000u) ? (2u << (2 * j)) : 0u;
        }
        mk[pix * (F >> 3) + (n >> 3)] = (unsigned char)bits;
      }
    }
  }
}

// one persistent block per CU over all groups
template <class G>
static int pconv_fwd_launch(const PConvFwdArgs& a, int ngroups, hipStream_t st, const char* name) {
  static bool opted[64] = {};
  static int ncu[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), name);
  SERL_CHECK_ARG(dev >= 0 && dev < 64, "%s: device ordinal %d out of range", name, dev);
  if (!opted[dev]) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)pconv_fwd_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS), name);
    SERL_CHECK_HIP(hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev), name);
    opted[dev] = true;
  }
  const int nitems = a.B * a.nbands;
  static int items_per_block = -1;
  if (items_per_block < 0) {
    const char* e = getenv("SERL_PCONV_ITEMS");
    items_per_block = e ? atoi(e) : 0;
  }
  int gx = pconv_cus(ncu[dev]) / ngroups;
  if (items_per_block > 0) gx = cdiv(nitems, items_per_block);
  if (gx < 1) gx = 1;
  if (gx > nitems) gx = nitems;
  hipLaunchKernelGGL((pconv_fwd_kernel<G>), dim3(gx, 1, ngroups), dim3(G::NT), G::LDS, st, a);
  SERL_CHECK_LAUNCH(name);
  return SERL_OK;
}

// ------------------------------------------------------------------------------------------------ forward, wide layers
// The same patch forward for a layer whose weight matrix no longer fits the registers of an 8-wave block (C = 64 ->
// F = 128: 9C x F x 2 planes = 295 KB): FOUR waves, one per SIMD, so that each may use the whole 512-register budget of
// its SIMD (hipcc spreads them over VGPRs and AGPRs; the MFMA B operand is read from either).  Wave w keeps the B
// fragments of output channels [32 w, 32 w + 32) for the whole K range (9C/16 x 8 = 288 registers) and multiplies ALL
// M tiles of the item by them (MT independent accumulators, which also keeps the MFMA pipe of its SIMD full without a
// second wave).  The im2col GEMM this replaces fetches every input line 2.25 times and streams the weights once per
// row tile; with the 4 MB L2 thrashed by the 32 blocks of an XCD both come from the Infinity Cache at ~6 TB/s (ablation:
// 155 us with real operands, 93 us when every load hits).  Here the input is read once (+ one halo row per band) and the
// weights once per persistent block.  Same k order and MFMA sequence per accumulator: bit-identical output.
template <int C_, int F_, int R_, int OWP_>
struct PFwdWCfg {
  static constexpr int C = C_, F = F_, R = R_, OWP = OWP_;
  static constexpr int NW = F_ / 32, NT = 64 * NW;
  static constexpr int REC = 4 * C, NCH = C / 4;
  static constexpr int NR = 2 * R + 1;
  static constexpr int ROWB = 2 * OWP * REC;
  static constexpr int PATCH = NR * ROWB;
  static constexpr int SLOTS = PATCH / 16, NDMA = SLOTS / NT;
  static constexpr int MT = (R * OWP + 31) / 32;
  static constexpr int KP = 9 * C, NS = KP / 16;
  static constexpr int EPI = MT * 32 * (F + 4) * 4;
  static constexpr int LDS = 2 * PATCH;
  static constexpr int CH = F / 8, RPP = NT / CH, PASSES = MT * 32 / RPP;
  static_assert(NW == 4, "one wave per SIMD");
  static_assert(C % 16 == 0 && 32 % OWP == 0 && (R * OWP) % 32 == 0 && SLOTS % NT == 0 && NDMA * 2 == NS, "shape");
  static_assert(EPI <= PATCH && LDS <= 160 * 1024, "LDS");
  static_assert(PASSES * 3 <= 63, "store count of the counted wait");
};

template <class G>
__global__ __launch_bounds__(G::NT, 1) void pconv_fwdw_kernel(const PConvFwdArgs a) {
  constexpr int C = G::C, F = G::F, R = G::R, OWP = G::OWP, NT = G::NT, NW = G::NW, MT = G::MT;
  constexpr int REC = G::REC, NCH = G::NCH, ROWB = G::ROWB, PARB = OWP * REC;
  constexpr int RPT = 32 / OWP;
  constexpr int KS = C / 16, NS = G::NS, Kp = G::KP;
  extern __shared__ __attribute__((aligned(16))) unsigned char p_smem[];

  const int z = blockIdx.z;
  const int nitems = a.B * a.nbands;
  const int per = (nitems + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it0 = blockIdx.x * per, it1 = min(nitems, it0 + per);
  if (it0 >= it1) return;                                   // block-uniform, before any barrier
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;

  constexpr int NDMA = G::NDMA;
  struct DmaSrc { const char* src; int nrows; };
  auto dma_src = [&](int item) {
    const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
    const int oy0 = band * R;
    DmaSrc d;
    d.nrows = 2 * min(R, a.OH - oy0) + 1;
    d.src = reinterpret_cast<const char*>(a.x[z]) + ((size_t)(b * a.IH + 2 * oy0) * a.IW) * REC;
    return d;
  };
  auto dma_piece = [&](const DmaSrc& d, unsigned char* patch, int i, bool on) {      // piece i of NDMA: NW waves x 64 slots
    const int l = (i * NW + wid) * 64 + lane;
    const int rec = l / NCH, chp = l & (NCH - 1);
    const int rr = rec / (2 * OWP), par = (rec / OWP) & 1, j = rec & (OWP - 1);
    const int ix = 2 * j + par;
#ifdef PCONVW_ZERO_SRC   // ablation build: every patch DMA reads the zero page
    const bool ok = false;
#else
    const bool ok = on && rr < d.nrows && ix < a.IW;
#endif
    const char* g = ok ? d.src + (size_t)(((rr * a.IW + ix) * NCH + (chp ^ pconv_swz<REC>(j))) * 16)
                       : reinterpret_cast<const char*>(g_zero_page);
    __builtin_amdgcn_global_load_lds((pconv_gptr_t)g, (pconv_lptr_t)(patch + (i * NW + wid) * 1024), 16, 0, 0);
  };
  {   // the first patch is on its way while the weights are loaded
    const DmaSrc d0 = dma_src(it0);
#pragma unroll 1
    for (int i = 0; i < NDMA; ++i) dma_piece(d0, p_smem, i, true);
  }

  // ---- weights: lane (r, h) holds k = 16 s + 8 h .. + 7 of output channel f = 32 wid + r for every k16 step s, straight
  // from the [F][2][Kp] layout (each lane walks its own row: once per persistent block)
  bf16x8_t rbh[NS], rbl[NS];
  {
#ifdef PCONVW_ZERO_W    // ablation build: weights from the zero page (64 lanes, 128 bytes)
    const unsigned short* wrow = reinterpret_cast<const unsigned short*>(g_zero_page) + h * 8 - (wid < 0 ? Kp : 0);
#else
    const unsigned short* wrow = a.wf[z] + (size_t)((wid * 32 + r) * 2) * Kp + h * 8;
#endif
    if (a.wfrag) {
      // round 3: serl_convs_prep_weights also writes the fragments in the order this loop wants them (72 KB per wave, one
      // contiguous span; walking the [F][2][Kp] rows made every lane fetch its own cache line: ~15 us per block)
      const u32x4_t* wq = reinterpret_cast<const u32x4_t*>(a.wf[z] + (size_t)2 * F * Kp) + (size_t)wid * NS * 128 + lane;
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        rbh[s] = __builtin_bit_cast(bf16x8_t, wq[s * 128]);
        rbl[s] = __builtin_bit_cast(bf16x8_t, wq[s * 128 + 64]);
      }
    } else {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        rbh[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(wrow + s * 16));
        rbl[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(wrow + Kp + s * 16));
      }
    }
  }

  // ---- fragment addresses of this lane (the same for every item)
  const int ox = r & (OWP - 1);
  int arow[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) arow[m] = 2 * min(m * RPT + r / OWP, R - 1) * ROWB;
  const int rec0 = ox * REC, rec1 = (ox + 1) * REC;
  const int sw0 = pconv_swz<REC>(ox) << 4, sw1 = pconv_swz<REC>(ox + 1) << 4;
  // The products are computed TRANSPOSED (weight fragment = the MFMA's A operand, pixel fragment = its B operand: the same
  // registers, the same products over the same k, the same bits): a lane then holds, for its OWN pixel, output channels
  // 32 wid + 8 g + 4 h .. + 3 in accumulator quad g -- the output leaves straight from the registers (below).
  float4 bsq[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bsq[g] = *reinterpret_cast<const float4*>(a.bias[z] + wid * 32 + 8 * g + 4 * h);
  unsigned short* const out = a.y[z];
  unsigned char* const mk = a.rmask[z];
  // every ordinary load of the block is done before the item loop: inside it only the DMA and the stores are counted
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  int pb = 0;
  for (int item = it0; item < it1; ++item, pb ^= 1) {
    unsigned char* const patch = p_smem + pb * G::PATCH;
    const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
    const int oy0 = band * R;
    const int rows_here = min(R, a.OH - oy0);
    // this item's patch has landed; the stores of the previous item are younger than its DMA and vmcnt retires in
    // order, so the wait leaves exactly those stores in flight (2, or 3 with a byte mask, per pass)
    if (item == it0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (mk != nullptr) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * G::PASSES) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * G::PASSES) : "memory");
    __builtin_amdgcn_s_barrier();
    const bool more = item + 1 < it1;
    const DmaSrc dn = dma_src(more ? item + 1 : item);
    unsigned char* const npatch = p_smem + (pb ^ 1) * G::PATCH;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[m][q] = 0.f;
    // One wave per SIMD: nobody else hides this wave's LDS latency, so the A fragments of k16 step s + 1 are requested
    // before the MFMAs of step s, and the accumulators of the M tiles alternate so that no MFMA waits for the one issued
    // just before it.  The reads are written by hand with a COUNTED wait (PF_LD_2X128): left to hipcc every wait was
    // lgkmcnt(0), i.e. every second step waited for the fragments it had just requested.
    static_assert(NS == 36 && KS == 4 && MT == 2 && RPT == 2 && R == 4, "K loop unrolled by hand for C = 64, two M tiles");
    u32x4_t fh[2][MT], fl[2][MT];
    uint32_t pa[2][KS][2];                 // [kx == 2][kc][plane]: lane address in this item's patch; taps and the M tile are immediates
    {
      const uint32_t pl = (uint32_t)(size_t)(pconv_lptr_t)patch + (uint32_t)arow[0];
#pragma unroll
      for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
        for (int kc = 0; kc < KS; ++kc)
#pragma unroll
          for (int p2 = 0; p2 < 2; ++p2)
            pa[x2][kc][p2] = pl + (uint32_t)((x2 ? rec1 : rec0) + (((kc * 2 + h + p2 * (C / 8)) << 4) ^ (x2 ? sw1 : sw0)));
    }
#define PW_IMM(s) (((s) / KS / 3) * ROWB + ((((s) / KS) % 3) & 1) * PARB)
#define PW_LD(buf, s)                                                                                                   \
    PF_LD_2X128(fh[buf][0], fl[buf][0], pa[((s) / KS) % 3 == 2][(s) % KS][0], pa[((s) / KS) % 3 == 2][(s) % KS][1], PW_IMM(s), PW_IMM(s)); \
    PF_LD_2X128(fh[buf][1], fl[buf][1], pa[((s) / KS) % 3 == 2][(s) % KS][0], pa[((s) / KS) % 3 == 2][(s) % KS][1],     \
                PW_IMM(s) + 4 * ROWB, PW_IMM(s) + 4 * ROWB);
#define PW_STEP(s, CNT)                                                                                                 \
    {                                                                                                                   \
      if ((s) + 1 < NS) { PW_LD(((s) + 1) & 1, ((s) + 1) % NS) }                                                        \
      asm volatile("s_waitcnt lgkmcnt(" #CNT ")"                                                                        \
                   : "+v"(fh[(s) & 1][0]), "+v"(fl[(s) & 1][0]), "+v"(fh[(s) & 1][1]), "+v"(fl[(s) & 1][1]) : : "memory"); \
      _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                    \
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rbh[s], __builtin_bit_cast(bf16x8_t, fl[(s) & 1][m]), acc[m], 0, 0, 0); \
      _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                    \
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rbl[s], __builtin_bit_cast(bf16x8_t, fh[(s) & 1][m]), acc[m], 0, 0, 0); \
      _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                                    \
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rbh[s], __builtin_bit_cast(bf16x8_t, fh[(s) & 1][m]), acc[m], 0, 0, 0); \
      /* the next item's DMA pieces behind the first NDMA / 2 k16 steps (early: the patch has the rest of the item to */ \
      /* land); branch-free (behind the last item the pieces copy zeros into the idle buffer) */                        \
      if (2 * (s) < NDMA) {                                                                                             \
        dma_piece(dn, npatch, 2 * (s), more);                                                                           \
        dma_piece(dn, npatch, 2 * (s) + 1, more);                                                                       \
      }                                                                                                                 \
      __builtin_amdgcn_sched_barrier(0);                                                                                \
    }
    PW_LD(0, 0)
    PW_STEP(0, 4) PW_STEP(1, 4) PW_STEP(2, 4) PW_STEP(3, 4) PW_STEP(4, 4) PW_STEP(5, 4) PW_STEP(6, 4) PW_STEP(7, 4) PW_STEP(8, 4)
    PW_STEP(9, 4) PW_STEP(10, 4) PW_STEP(11, 4) PW_STEP(12, 4) PW_STEP(13, 4) PW_STEP(14, 4) PW_STEP(15, 4) PW_STEP(16, 4) PW_STEP(17, 4)
    PW_STEP(18, 4) PW_STEP(19, 4) PW_STEP(20, 4) PW_STEP(21, 4) PW_STEP(22, 4) PW_STEP(23, 4) PW_STEP(24, 4) PW_STEP(25, 4) PW_STEP(26, 4)
    PW_STEP(27, 4) PW_STEP(28, 4) PW_STEP(29, 4) PW_STEP(30, 4) PW_STEP(31, 4) PW_STEP(32, 4) PW_STEP(33, 4) PW_STEP(34, 4) PW_STEP(35, 0)
#undef PW_STEP
#undef PW_LD
#undef PW_IMM
    // ---- S-format output straight from the accumulators (no LDS tile, no barrier).  Per pixel row a lane pair (r, h = 0 / 1)
    // holds channel quads 8 g + 4 h .. + 3; after bias / ReLU / hi-lo split one v_permlane32_swap per dword and quad pair
    // (vdst = quad k, src = quad k + 1: lanes 32-63 of vdst swap with lanes 0-31 of src) leaves 8 consecutive channels in
    // every lane -- 16-byte stores of the hi and of the lo plane, and the mask byte of those 8 channels.  Every lane issues
    // every store (the wait at the top of the next item is a COUNT): lanes whose pixel does not exist (the padding column,
    // rows past the image's last band) store to a dump area nobody reads.
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int row0 = m * 32 + r;
      const int oyl = row0 / OWP, oxx = row0 & (OWP - 1);
      const bool valid = oyl < rows_here && oxx < a.OW;
      const size_t pix = (size_t)(b * a.OH + oy0 + (valid ? oyl : 0)) * a.OW + (valid ? oxx : 0);
      uint32_t hq[4][2], lq[4][2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v0 = fmaf(acc[m][4 * g], 1.0f, bsq[g].x), v1 = fmaf(acc[m][4 * g + 1], 1.0f, bsq[g].y);
        float v2 = fmaf(acc[m][4 * g + 2], 1.0f, bsq[g].z), v3 = fmaf(acc[m][4 * g + 3], 1.0f, bsq[g].w);
        if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        split_pair(v0, v1, hq[g][0], lq[g][0]);
        split_pair(v2, v3, hq[g][1], lq[g][1]);
      }
#pragma unroll
      for (int k = 0; k < 4; k += 2) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          auto sh = __builtin_amdgcn_permlane32_swap(hq[k][e], hq[k + 1][e], false, false);
          hq[k][e] = sh[0]; hq[k + 1][e] = sh[1];
          auto sl = __builtin_amdgcn_permlane32_swap(lq[k][e], lq[k + 1][e], false, false);
          lq[k][e] = sl[0]; lq[k + 1][e] = sl[1];
        }
        const u32x4_t hi = {hq[k][0], hq[k][1], hq[k + 1][0], hq[k + 1][1]}, lo = {lq[k][0], lq[k][1], lq[k + 1][0], lq[k + 1][1]};
        const int chan = wid * 32 + 8 * k + 8 * h;          // lanes 0-31: channels of quads k (both halves); lanes 32-63: of quad k + 1
        unsigned short* dst = valid ? out + pix * 2 * F + chan : reinterpret_cast<unsigned short*>(g_store_dump) + lane * 8;
        *reinterpret_cast<u32x4_t*>(dst) = hi;
        *reinterpret_cast<u32x4_t*>(dst + (valid ? F : 512)) = lo;
        if (mk != nullptr) {
          uint32_t bits = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bits |= (hi[j] & 0x7fffu) ? (1u << (2 * j)) : 0u;
            bits |= (hi[j] & 0x7fff0000u) ? (2u << (2 * j)) : 0u;
          }
          unsigned char* md = valid ? mk + pix * (F >> 3) + (chan >> 3) : reinterpret_cast<unsigned char*>(g_store_dump) + 2048 + lane;
          *md = (unsigned char)bits;
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may still be landing when the block's LDS is handed on
}

template <class G>
static int pconv_fwdw_launch(const PConvFwdArgs& a, int ngroups, hipStream_t st, const char* name) {
  static bool opted[64] = {};
  static int ncu[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), name);
  SERL_CHECK_ARG(dev >= 0 && dev < 64, "%s: device ordinal %d out of range", name, dev);
  if (!opted[dev]) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)pconv_fwdw_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS), name);
    SERL_CHECK_HIP(hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev), name);
    opted[dev] = true;
  }
  const int nitems = a.B * a.nbands;
  static int cap = -1;                    // SERL_PCONV_FWDW_CUS: CUs this kernel spreads over (experiments with co-running streams)
  if (cap < 0) { const char* e = getenv("SERL_PCONV_FWDW_CUS"); cap = e ? atoi(e) : 0; }
  const int cus = pconv_cus(ncu[dev]);
  int gx = (cap > 0 && cap < cus ? cap : cus) / ngroups;
  if (gx < 1) gx = 1;
  if (gx > nitems) gx = nitems;
  gx = cdiv(nitems, cdiv(nitems, gx));
  hipLaunchKernelGGL((pconv_fwdw_kernel<G>), dim3(gx, 1, ngroups), dim3(G::NT), G::LDS, st, a);
  SERL_CHECK_LAUNCH(name);
  return SERL_OK;
}

// ------------------------------------------------------------------------------------------------ data gradient
// dx = conv_transpose(dz, w) for a 3x3 stride-2 layer with C = 32, F = 64, masked by the ReLU byte mask of x.
// Per parity class (py, px) of input pixels the taps with ky = py (mod 2), kx = px (mod 2) contribute:
//   dx[2 iyp + py][2 ixp + px][c] = sum_{jy, jx, f} dz[iyp - jy][ixp - jx][f] * w[py + 2 jy][px + 2 jx][c][f]
// (4 / 2 / 2 / 1 taps), the same k order (jy, jx, f) as sgemm_kernel<ConvDgradS>: bit-identical output.
// The im2col kernel runs the four classes as separate GEMMs that each gather dz from global memory (measured 1.65x
// the algorithmic bytes).  Here an item is a band of P = 4 input row PAIRS of one image: its P + 1 rows of dz are
// staged once in LDS by LDS-DMA (double-buffered; record 31 of every row and the rows outside the image are zeros, so
// out-of-range taps need no predicate) and all four classes read their A fragments from that patch.
// 8 waves: wave 2p handles class (0,0) of row pair p (16 k16 steps), wave 2p+1 classes (0,1), (1,0), (1,1) (8 + 8 + 4):
// each keeps exactly its classes' weight fragments in registers (128 / 160 VGPRs).  The output leaves through a
// wave-private fp32 tile in LDS, so an item needs one workgroup barrier.
struct PConvDgradArgs {
  const unsigned short* dz[MAXG];
  const unsigned short* wd[MAXG];     // S format [(ky, kx, c)][2][F]
  const unsigned char* xbits[MAXG];   // byte mask [B*IH*IW][C/8] or nullptr
  unsigned short* dx[MAXG];
  int B, IH, IW, OH, OW, nbands;
  FastDiv d_bands;
};

__device__ uint4 g_pconv_dump[320];     // stores of lanes that hold no valid pixel land here (never read)

struct PDgradCfg {
  static constexpr int C = 32, F = 64, P = 4, NT = 512;
  static constexpr int REC = 4 * F, NCH = F / 4;          // 256-byte records, 16 chunks
  static constexpr int ROWB = 32 * REC;
  static constexpr int PATCH = (P + 1) * ROWB;
  static constexpr int SLOTS = PATCH / 16, NDMA = SLOTS / NT;
  static constexpr int CSLD = C + 4, CSB = 32 * CSLD * 4;  // wave-private output tile
  static constexpr int MSB = 2 * P * 64 * (C / 8);        // mask bytes of one item: 2P rows x (IW <= 64) pixels x C/8
  static constexpr int LDS = 2 * PATCH + 8 * CSB + 2 * MSB;
  static_assert(SLOTS % NT == 0 && MSB == 8 * 256 && LDS <= 160 * 1024, "LDS");
};

// K loop of one 32-pixel tile of class (PY, PX): row pair iyl of the band, B fragments bh/bl[t * 4 + kc]
template <int PY, int PX>
__device__ __forceinline__ void pdgrad_tile(const unsigned char* patch, int iyl, int r, int h, const bf16x8_t* bh,
                                            const bf16x8_t* bl, f32x16& acc) {
  constexpr int TY = PY ? 1 : 2, TX = PX ? 1 : 2;
  constexpr int ROWB = PDgradCfg::ROWB;
#pragma unroll
  for (int jy = 0; jy < TY; ++jy)
#pragma unroll
    for (int jx = 0; jx < TX; ++jx) {
      const int t = jy * TX + jx;
      const int rec = (r - jx) & 31;
      const unsigned char* base = patch + (iyl - jy + 1) * ROWB + rec * 256;
      const int sw = (rec & 15) << 4;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(base + (((kc * 2 + h) << 4) ^ sw)));
        const bf16x8_t al = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(base + (((kc * 2 + h + 8) << 4) ^ sw)));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[t * 4 + kc], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[t * 4 + kc], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[t * 4 + kc], acc, 0, 0, 0);
      }
    }
}

// B fragments of class (PY, PX): lane (r = c, h) holds f = 16 kc + 8 h .. + 7 of tap t
template <int PY, int PX>
__device__ __forceinline__ void pdgrad_load_b(const unsigned short* wd, int r, int h, bf16x8_t* bh, bf16x8_t* bl) {
  constexpr int TY = PY ? 1 : 2, TX = PX ? 1 : 2, C = PDgradCfg::C, F = PDgradCfg::F;
#pragma unroll
  for (int jy = 0; jy < TY; ++jy)
#pragma unroll
    for (int jx = 0; jx < TX; ++jx) {
      const int tap = (PY + 2 * jy) * 3 + (PX + 2 * jx);
      const unsigned short* row = wd + (size_t)((tap * C + r) * 2) * F + h * 8;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        bh[(jy * TX + jx) * 4 + kc] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(row + kc * 16));
        bl[(jy * TX + jx) * 4 + kc] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(row + F + kc * 16));
      }
    }
}

// One tile out: accumulators -> wave-private fp32 tile -> per lane two (pixel, 8-channel) units: ReLU byte mask (from
// the mask bytes the DMA brought into LDS at ms_base, or none), hi / lo split, two 16-byte stores.  Lanes whose pixel is
// outside the image store to g_pconv_dump, so that every wave issues the same number of stores per item.
// LDS reads by hand: hipcc puts vmcnt(0) in front of an LDS read it cannot tell apart from an LDS-DMA destination (and
// in front of the use of any ordinary global load while a DMA is in flight), which would wait for the NEXT item's patch.
template <int PY, int PX>
__device__ __forceinline__ void pdgrad_store(const PConvDgradArgs& a, int z, int b, int band, int iyl, float* Cs,
                                             uint32_t cs_base, uint32_t ms_base, bool masked, const f32x16& acc, int lane) {
  constexpr int C = PDgradCfg::C, LD = PDgradCfg::CSLD, P = PDgradCfg::P;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 16; ++q) Cs[((q & 3) + 8 * (q >> 2) + 4 * h) * LD + r] = acc[q];
  const int iy = 2 * (band * P + iyl) + PY;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int u = lane + 64 * pass, pxl = u >> 2, ch = u & 3;
    const int ix = 2 * pxl + PX;
    const bool ok = iy < a.IH && ix < a.IW;
    u32x4_t c0, c1;
    uint32_t bits = 0xffu;
    if (masked) {
      const uint32_t maddr = ms_base + (ok ? (uint32_t)(((2 * iyl + PY) * a.IW + ix) * (C >> 3) + ch) : 0u);
      asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_u8 %2, %4\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(c0), "=&v"(c1), "=&v"(bits) : "v"(cs_base + (uint32_t)(pxl * LD + ch * 8) * 4u), "v"(maddr) : "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(c0), "=&v"(c1) : "v"(cs_base + (uint32_t)(pxl * LD + ch * 8) * 4u) : "memory");
    }
    float v[8] = {__uint_as_float(c0[0]), __uint_as_float(c0[1]), __uint_as_float(c0[2]), __uint_as_float(c0[3]),
                  __uint_as_float(c1[0]), __uint_as_float(c1[1]), __uint_as_float(c1[2]), __uint_as_float(c1[3])};
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (!(bits & (1u << j))) v[j] = 0.f;
    u32x4_t hi, lo;
    uint32_t hh, ll;
    split_pair(v[0], v[1], hh, ll); hi[0] = hh; lo[0] = ll;
    split_pair(v[2], v[3], hh, ll); hi[1] = hh; lo[1] = ll;
    split_pair(v[4], v[5], hh, ll); hi[2] = hh; lo[2] = ll;
    split_pair(v[6], v[7], hh, ll); hi[3] = hh; lo[3] = ll;
    unsigned short* dst = ok ? a.dx[z] + ((size_t)(b * a.IH + iy) * a.IW + ix) * 2 * C + ch * 8
                             : reinterpret_cast<unsigned short*>(g_pconv_dump) + lane * 32;
    *reinterpret_cast<u32x4_t*>(dst) = hi;
    *reinterpret_cast<u32x4_t*>(dst + C) = lo;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the tile is reused by this wave's next class
}

__global__ __launch_bounds__(512, 1) void pconv_dgrad_kernel(const PConvDgradArgs a) {
  typedef PDgradCfg G;
  constexpr int P = G::P, NT = G::NT, ROWB = G::ROWB, NDMA = G::NDMA;
  extern __shared__ __attribute__((aligned(16))) unsigned char p_smem[];
  const int z = blockIdx.z;
  const int nitems = a.B * a.nbands;
  const int per = (nitems + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it0 = blockIdx.x * per, it1 = min(nitems, it0 + per);
  if (it0 >= it1) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  // waves w and w + 4 share a SIMD: one wave of each type per SIMD (48 + 60 MFMAs per item on each of the four; with
  // the types alternating by wave the SIMDs got 96 or 120: 120.6 -> 118.5 us stand-alone)
  const int iyl = wid & 3;                  // row pair of the band
  const bool type_x = wid < 4;              // class (0,0) | classes (0,1), (1,0), (1,1)
  float* const Cs = reinterpret_cast<float*>(p_smem + 2 * G::PATCH + wid * G::CSB);
  const uint32_t cs_base = (uint32_t)(size_t)(pconv_lptr_t)Cs;

  unsigned char* const Ms = p_smem + 2 * G::PATCH + 8 * G::CSB;          // [2][MSB] mask bytes of the item's 2P rows
  const uint32_t ms_base0 = (uint32_t)(size_t)(pconv_lptr_t)Ms;
  const bool masked = a.xbits[z] != nullptr;

  struct DmaSrc { const char* dz; const char* mk; int oy_first, mk_bytes; };
  auto dma_src = [&](int item) {
    const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
    DmaSrc d;
    d.oy_first = band * P - 1;              // dz row of patch row 0
    d.dz = reinterpret_cast<const char*>(a.dz[z]) + (size_t)b * a.OH * a.OW * G::REC;
    // the mask bytes of input rows 2P*band .. + 2P - 1 are one contiguous span
    d.mk = reinterpret_cast<const char*>(a.xbits[z]) + ((size_t)(b * a.IH + 2 * P * band) * a.IW) * (G::C / 8);
    d.mk_bytes = max(0, min(2 * P, a.IH - 2 * P * band)) * a.IW * (G::C / 8);
    return d;
  };
  auto dma_mask = [&](const DmaSrc& d, int buf) {           // wave w brings bytes [256 w, 256 w + 256) of the span
    if (!masked) return;
    const int off = wid * 256 + lane * 4;
    const char* g = off < d.mk_bytes ? d.mk + off : reinterpret_cast<const char*>(g_zero_page);
    __builtin_amdgcn_global_load_lds((pconv_gptr_t)g, (pconv_lptr_t)(Ms + buf * G::MSB + wid * 256), 4, 0, 0);
  };
  auto dma_piece = [&](const DmaSrc& d, unsigned char* patch, int i) {
    const int l = (i * 8 + wid) * 64 + lane;
    const int rec = l >> 4, chp = l & 15;
    const int row = rec >> 5, ox = rec & 31, oy = d.oy_first + row;
    const bool ok = oy >= 0 && oy < a.OH && ox < a.OW;
    const char* g = ok ? d.dz + (size_t)(((oy * a.OW + ox) << 4) + (chp ^ (ox & 15))) * 16
                       : reinterpret_cast<const char*>(g_zero_page);
    __builtin_amdgcn_global_load_lds((pconv_gptr_t)g, (pconv_lptr_t)(patch + (i * 8 + wid) * 1024), 16, 0, 0);
  };
  {
    const DmaSrc d0 = dma_src(it0);
#pragma unroll 1
    for (int i = 0; i < NDMA; ++i) dma_piece(d0, p_smem, i);
    dma_mask(d0, 0);
  }

  if (type_x) {
    bf16x8_t bh[16], bl[16];
    pdgrad_load_b<0, 0>(a.wd[z], r, h, bh, bl);
    int pb = 0;
    for (int item = it0; item < it1; ++item, pb ^= 1) {
      const unsigned char* patch = p_smem + pb * G::PATCH;
      unsigned char* npatch = p_smem + (pb ^ 1) * G::PATCH;
      const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
      // this item's patch has landed; the stores of the previous item (4 per wave of this type) are younger than its
      // DMA and vmcnt retires in order, so they stay in flight
      if (item == it0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const bool more = item + 1 < it1;
      const DmaSrc dn = dma_src(more ? item + 1 : item);
      if (more) {
#pragma unroll
        for (int i = 0; i < NDMA; ++i) dma_piece(dn, npatch, i);
        dma_mask(dn, pb ^ 1);
      }
      const uint32_t ms_base = ms_base0 + pb * G::MSB;
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      pdgrad_tile<0, 0>(patch, iyl, r, h, bh, bl, acc);
      pdgrad_store<0, 0>(a, z, b, band, iyl, Cs, cs_base, ms_base, masked, acc, lane);
    }
  } else {
    bf16x8_t bh[20], bl[20];
    pdgrad_load_b<0, 1>(a.wd[z], r, h, bh, bl);
    pdgrad_load_b<1, 0>(a.wd[z], r, h, bh + 8, bl + 8);
    pdgrad_load_b<1, 1>(a.wd[z], r, h, bh + 16, bl + 16);
    int pb = 0;
    for (int item = it0; item < it1; ++item, pb ^= 1) {
      const unsigned char* patch = p_smem + pb * G::PATCH;
      unsigned char* npatch = p_smem + (pb ^ 1) * G::PATCH;
      const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
      if (item == it0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");      // 3 classes x 4 stores
      __builtin_amdgcn_s_barrier();
      const bool more = item + 1 < it1;
      const DmaSrc dn = dma_src(more ? item + 1 : item);
      if (more) {
#pragma unroll
        for (int i = 0; i < NDMA; ++i) dma_piece(dn, npatch, i);
        dma_mask(dn, pb ^ 1);
      }
      const uint32_t ms_base = ms_base0 + pb * G::MSB;
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      pdgrad_tile<0, 1>(patch, iyl, r, h, bh, bl, acc);
      pdgrad_store<0, 1>(a, z, b, band, iyl, Cs, cs_base, ms_base, masked, acc, lane);
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      pdgrad_tile<1, 0>(patch, iyl, r, h, bh + 8, bl + 8, acc);
      pdgrad_store<1, 0>(a, z, b, band, iyl, Cs, cs_base, ms_base, masked, acc, lane);
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      pdgrad_tile<1, 1>(patch, iyl, r, h, bh + 16, bl + 16, acc);
      pdgrad_store<1, 1>(a, z, b, band, iyl, Cs, cs_base, ms_base, masked, acc, lane);
    }
  }
}

static int pconv_dgrad_launch(const PConvDgradArgs& a, int ngroups, hipStream_t st, const char* name) {
  static bool opted[64] = {};
  static int ncu[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), name);
  SERL_CHECK_ARG(dev >= 0 && dev < 64, "%s: device ordinal %d out of range", name, dev);
  if (!opted[dev]) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)pconv_dgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PDgradCfg::LDS), name);
    SERL_CHECK_HIP(hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev), name);
    opted[dev] = true;
  }
  const int nitems = a.B * a.nbands;
  static int cap = -1;                    // SERL_PCONV_DG_CUS: CUs this kernel spreads over (experiments with co-running streams)
  if (cap < 0) { const char* e = getenv("SERL_PCONV_DG_CUS"); cap = e ? atoi(e) : 0; }
  const int cus = pconv_cus(ncu[dev]);
  int gx = (cap > 0 && cap < cus ? cap : cus) / ngroups;
  if (gx < 1) gx = 1;
  if (gx > nitems) gx = nitems;
  gx = cdiv(nitems, cdiv(nitems, gx));
  hipLaunchKernelGGL(pconv_dgrad_kernel, dim3(gx, 1, ngroups), dim3(PDgradCfg::NT), PDgradCfg::LDS, st, a);
  SERL_CHECK_LAUNCH(name);
  return SERL_OK;
}

// ------------------------------------------------------------------------------------------------ data gradient, wide layer
// The same data gradient for C = 64, F = 128 (third layer of the encoder: dz 15 x 15 x 128 -> dx 31 x 31 x 64).  The
// im2col kernel fetched 4.7x the bytes of dz here (profiles/r02_v4_traffic.json).  The weights of all four parity classes
// (9 taps x 64 x 128 x 2 planes = 295 KB) do not fit the registers of an 8-wave block, so -- as in pconv_fwdw_kernel --
// FOUR waves, one per SIMD, each with the SIMD's whole register budget: wave (nh, cg) owns output channels
// [32 nh, 32 nh + 32) of class (0,0) (cg = 0: 4 taps, K = 512, 256 registers of B fragments) or of classes (0,1), (1,0),
// (1,1) (cg = 1: K = 256 + 256 + 128, 320 registers): 32 : 40 k16 steps per M tile.  An item is a band of P = 4 input
// row pairs of one image (P + 1 rows of dz, 512-byte records, 16 per row: record 15 and the rows outside the image
// are zeros); per class it has 4 x 16 pixels = two M tiles (tile m: row pairs 2m, 2m + 1; lane r: column pair r & 15 of
// row pair 2m + (r >> 4)), multiplied together with alternating accumulators and the A fragments one k16 step ahead.
// Same k order (jy, jx, f) and MFMA sequence per output as sgemm_kernel<ConvDgradS>: bit-identical.
struct PDgradWCfg {
  static constexpr int C = 64, F = 128, P = 4, NW = 4, NT = 64 * NW;
  static constexpr int REC = 4 * F, NCH = F / 4;          // 512-byte records, 32 chunks
  static constexpr int ROWB = 16 * REC;
  static constexpr int PATCH = (P + 1) * ROWB;
  static constexpr int SLOTS = PATCH / 16, NDMA = SLOTS / NT;
  static constexpr int CSLD = 32 + 4, CSB = 32 * CSLD * 4;  // wave-private output tile (32 pixels x 32 channels)
  static constexpr int MSB = 2 * P * 32 * (C / 8);        // mask bytes of one item: 2P rows x (IW <= 32) pixels x C/8
  static constexpr int LDS = 2 * PATCH + NW * CSB + 2 * MSB;
  static_assert(SLOTS % NT == 0 && MSB == 2 * NW * 256 && LDS <= 160 * 1024, "LDS");
};

// B fragments of class (PY, PX) for output channels [32 nh, 32 nh + 32): lane (r, h) holds f = 16 kc + 8 h .. + 7 of tap t
template <int PY, int PX>
__device__ __forceinline__ void pdgradw_load_b(const unsigned short* wd, int nh, int r, int h, bf16x8_t* bh, bf16x8_t* bl) {
  constexpr int TY = PY ? 1 : 2, TX = PX ? 1 : 2, C = PDgradWCfg::C, F = PDgradWCfg::F, KC = F / 16;
#pragma unroll
  for (int jy = 0; jy < TY; ++jy)
#pragma unroll
    for (int jx = 0; jx < TX; ++jx) {
      const int tap = (PY + 2 * jy) * 3 + (PX + 2 * jx);
      const unsigned short* row = wd + (size_t)((tap * C + nh * 32 + r) * 2) * F + h * 8;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
        bh[(jy * TX + jx) * KC + kc] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(row + kc * 16));
        bl[(jy * TX + jx) * KC + kc] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(row + F + kc * 16));
      }
    }
}

// K loop of both M tiles of class (PY, PX)
template <int PY, int PX>
__device__ __forceinline__ void pdgradw_tiles(const unsigned char* patch, int r, int h, const bf16x8_t* bh,
                                              const bf16x8_t* bl, f32x16 (&acc)[2]) {
  constexpr int TY = PY ? 1 : 2, TX = PX ? 1 : 2, KC = PDgradWCfg::F / 16, NSTEP = TY * TX * KC;
  constexpr int ROWB = PDgradWCfg::ROWB, REC = PDgradWCfg::REC;
  const int ixp = r & 15, rp = r >> 4;
  bf16x8_t fh[2][2], fl[2][2];
  auto frags = [&](int s, int buf) {
    const int t = s / KC, kc = s - t * KC;
    const int jy = t / TX, jx = t - jy * TX;
    const int rec = (ixp - jx) & 15;
    const unsigned char* base = patch + (rp - jy + 1) * ROWB + rec * REC;
    const int sw = rec << 4;
    const int chh = ((kc * 2 + h) << 4) ^ sw, chl = ((kc * 2 + h + PDgradWCfg::F / 8) << 4) ^ sw;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      fh[buf][m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(base + m * 2 * ROWB + chh));
      fl[buf][m] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(base + m * 2 * ROWB + chl));
    }
  };
  frags(0, 0);
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int cb = s & 1;
    if (s + 1 < NSTEP) frags(s + 1, cb ^ 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl[cb][m], bh[s], acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[cb][m], bl[s], acc[m], 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh[cb][m], bh[s], acc[m], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// M tile mt of class (PY, PX) out (pdgrad_store's path: wave-private fp32 tile, byte mask from LDS, two 16-byte stores
// per pass, lanes without a valid pixel store to g_pconv_dump so that every wave issues the same number of stores)
template <int PY, int PX>
__device__ __forceinline__ void pdgradw_store(const PConvDgradArgs& a, int z, int b, int band, int mt, int nh, float* Cs,
                                              uint32_t cs_base, uint32_t ms_base, bool masked, const f32x16& acc, int lane) {
  constexpr int C = PDgradWCfg::C, LD = PDgradWCfg::CSLD, P = PDgradWCfg::P;
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int q = 0; q < 16; ++q) Cs[((q & 3) + 8 * (q >> 2) + 4 * h) * LD + r] = acc[q];
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int u = lane + 64 * pass, pxl = u >> 2, ch = u & 3;
    const int rp = 2 * mt + (pxl >> 4);
    const int iy = 2 * (band * P + rp) + PY, ix = 2 * (pxl & 15) + PX;
    const bool ok = iy < a.IH && ix < a.IW;
    u32x4_t c0, c1;
    uint32_t bits = 0xffu;
    if (masked) {
      const uint32_t maddr = ms_base + (ok ? (uint32_t)(((2 * rp + PY) * a.IW + ix) * (C >> 3) + nh * 4 + ch) : 0u);
      asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_u8 %2, %4\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(c0), "=&v"(c1), "=&v"(bits) : "v"(cs_base + (uint32_t)(pxl * LD + ch * 8) * 4u), "v"(maddr) : "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(c0), "=&v"(c1) : "v"(cs_base + (uint32_t)(pxl * LD + ch * 8) * 4u) : "memory");
    }
    float v[8] = {__uint_as_float(c0[0]), __uint_as_float(c0[1]), __uint_as_float(c0[2]), __uint_as_float(c0[3]),
                  __uint_as_float(c1[0]), __uint_as_float(c1[1]), __uint_as_float(c1[2]), __uint_as_float(c1[3])};
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (!(bits & (1u << j))) v[j] = 0.f;
    u32x4_t hi, lo;
    uint32_t hh, ll;
    split_pair(v[0], v[1], hh, ll); hi[0] = hh; lo[0] = ll;
    split_pair(v[2], v[3], hh, ll); hi[1] = hh; lo[1] = ll;
    split_pair(v[4], v[5], hh, ll); hi[2] = hh; lo[2] = ll;
    split_pair(v[6], v[7], hh, ll); hi[3] = hh; lo[3] = ll;
    unsigned short* dst = ok ? a.dx[z] + ((size_t)(b * a.IH + iy) * a.IW + ix) * 2 * C + nh * 32 + ch * 8
                             : reinterpret_cast<unsigned short*>(g_pconv_dump) + lane * 32;
    *reinterpret_cast<u32x4_t*>(dst) = hi;
    *reinterpret_cast<u32x4_t*>(dst + C) = lo;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the tile is reused by this wave's next M tile
}

__global__ __launch_bounds__(256, 1) void pconv_dgradw_kernel(const PConvDgradArgs a) {
  typedef PDgradWCfg G;
  constexpr int P = G::P, NW = G::NW, NDMA = G::NDMA;
  extern __shared__ __attribute__((aligned(16))) unsigned char p_smem[];
  const int z = blockIdx.z;
  const int nitems = a.B * a.nbands;
  const int per = (nitems + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it0 = blockIdx.x * per, it1 = min(nitems, it0 + per);
  if (it0 >= it1) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nh = wid & 1;                   // half of the output channels
  const bool type_x = (wid >> 1) == 0;      // class (0,0) | classes (0,1), (1,0), (1,1)
  float* const Cs = reinterpret_cast<float*>(p_smem + 2 * G::PATCH + wid * G::CSB);
  const uint32_t cs_base = (uint32_t)(size_t)(pconv_lptr_t)Cs;
  unsigned char* const Ms = p_smem + 2 * G::PATCH + NW * G::CSB;         // [2][MSB] mask bytes of the item's 2P rows
  const uint32_t ms_base0 = (uint32_t)(size_t)(pconv_lptr_t)Ms;
  const bool masked = a.xbits[z] != nullptr;

  struct DmaSrc { const char* dz; const char* mk; int oy_first, mk_bytes; };
  auto dma_src = [&](int item) {
    const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
    DmaSrc d;
    d.oy_first = band * P - 1;              // dz row of patch row 0
    d.dz = reinterpret_cast<const char*>(a.dz[z]) + (size_t)b * a.OH * a.OW * G::REC;
    d.mk = reinterpret_cast<const char*>(a.xbits[z]) + ((size_t)(b * a.IH + 2 * P * band) * a.IW) * (G::C / 8);
    d.mk_bytes = max(0, min(2 * P, a.IH - 2 * P * band)) * a.IW * (G::C / 8);
    return d;
  };
  auto dma_mask = [&](const DmaSrc& d, int buf) {           // two rounds of 4 waves x 256 bytes
    if (!masked) return;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = (i * NW + wid) * 256 + lane * 4;
      const char* g = off < d.mk_bytes ? d.mk + off : reinterpret_cast<const char*>(g_zero_page);
      __builtin_amdgcn_global_load_lds((pconv_gptr_t)g, (pconv_lptr_t)(Ms + buf * G::MSB + (i * NW + wid) * 256), 4, 0, 0);
    }
  };
  auto dma_piece = [&](const DmaSrc& d, unsigned char* patch, int i) {
    const int l = (i * NW + wid) * 64 + lane;
    const int rec = l >> 5, chp = l & 31;
    const int row = rec >> 4, ox = rec & 15, oy = d.oy_first + row;
    const bool ok = oy >= 0 && oy < a.OH && ox < a.OW;
    const char* g = ok ? d.dz + (size_t)(((oy * a.OW + ox) << 5) + (chp ^ ox)) * 16
                       : reinterpret_cast<const char*>(g_zero_page);
    __builtin_amdgcn_global_load_lds((pconv_gptr_t)g, (pconv_lptr_t)(patch + (i * NW + wid) * 1024), 16, 0, 0);
  };
  {
    const DmaSrc d0 = dma_src(it0);
#pragma unroll 1
    for (int i = 0; i < NDMA; ++i) dma_piece(d0, p_smem, i);
    dma_mask(d0, 0);
  }

  if (type_x) {
    bf16x8_t bh[32], bl[32];
    pdgradw_load_b<0, 0>(a.wd[z], nh, r, h, bh, bl);
    int pb = 0;
    for (int item = it0; item < it1; ++item, pb ^= 1) {
      const unsigned char* patch = p_smem + pb * G::PATCH;
      unsigned char* npatch = p_smem + (pb ^ 1) * G::PATCH;
      const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
      // this item's patch has landed; the stores of the previous item (2 M tiles x 4 per wave of this type) are younger
      // than its DMA and vmcnt retires in order, so they stay in flight
      if (item == it0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const bool more = item + 1 < it1;
      const DmaSrc dn = dma_src(more ? item + 1 : item);
      if (more) {
#pragma unroll
        for (int i = 0; i < NDMA; ++i) dma_piece(dn, npatch, i);
        dma_mask(dn, pb ^ 1);
      }
      const uint32_t ms_base = ms_base0 + pb * G::MSB;
      f32x16 acc[2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[m][q] = 0.f;
      pdgradw_tiles<0, 0>(patch, r, h, bh, bl, acc);
      pdgradw_store<0, 0>(a, z, b, band, 0, nh, Cs, cs_base, ms_base, masked, acc[0], lane);
      pdgradw_store<0, 0>(a, z, b, band, 1, nh, Cs, cs_base, ms_base, masked, acc[1], lane);
    }
  } else {
    bf16x8_t bh[40], bl[40];
    pdgradw_load_b<0, 1>(a.wd[z], nh, r, h, bh, bl);
    pdgradw_load_b<1, 0>(a.wd[z], nh, r, h, bh + 16, bl + 16);
    pdgradw_load_b<1, 1>(a.wd[z], nh, r, h, bh + 32, bl + 32);
    int pb = 0;
    for (int item = it0; item < it1; ++item, pb ^= 1) {
      const unsigned char* patch = p_smem + pb * G::PATCH;
      unsigned char* npatch = p_smem + (pb ^ 1) * G::PATCH;
      const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
      if (item == it0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");      // 3 classes x 2 M tiles x 4 stores
      __builtin_amdgcn_s_barrier();
      const bool more = item + 1 < it1;
      const DmaSrc dn = dma_src(more ? item + 1 : item);
      if (more) {
#pragma unroll
        for (int i = 0; i < NDMA; ++i) dma_piece(dn, npatch, i);
        dma_mask(dn, pb ^ 1);
      }
      const uint32_t ms_base = ms_base0 + pb * G::MSB;
      f32x16 acc[2];
#define PDGW_CLASS(PY, PX, OFF)                                                                                         \
      _Pragma("unroll") for (int m = 0; m < 2; ++m)                                                                     \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) acc[m][q] = 0.f;                                                 \
      pdgradw_tiles<PY, PX>(patch, r, h, bh + OFF, bl + OFF, acc);                                                      \
      pdgradw_store<PY, PX>(a, z, b, band, 0, nh, Cs, cs_base, ms_base, masked, acc[0], lane);                          \
      pdgradw_store<PY, PX>(a, z, b, band, 1, nh, Cs, cs_base, ms_base, masked, acc[1], lane);
      PDGW_CLASS(0, 1, 0)
      PDGW_CLASS(1, 0, 16)
      PDGW_CLASS(1, 1, 32)
#undef PDGW_CLASS
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no DMA may still be landing when the block's LDS is handed on
}

static int pconv_dgradw_launch(const PConvDgradArgs& a, int ngroups, hipStream_t st, const char* name) {
  static bool opted[64] = {};
  static int ncu[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), name);
  SERL_CHECK_ARG(dev >= 0 && dev < 64, "%s: device ordinal %d out of range", name, dev);
  if (!opted[dev]) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)pconv_dgradw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PDgradWCfg::LDS), name);
    SERL_CHECK_HIP(hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev), name);
    opted[dev] = true;
  }
  const int nitems = a.B * a.nbands;
  // A block of this kernel takes its CU's whole register file, so on every CU it runs on the other streams of the
  // update (weight gradients, the actor's run-ahead pass) stand still: three quarters of the CUs measured best inside
  // the update (588 -> 591 grad-steps/s; all CUs: 584, i.e. slower than the im2col kernel it replaces, although it is
  // 15 % faster alone).  SERL_PCONV_DGW_CUS overrides the count.  Blocks get equal item counts.
  static int cap = -1;
  if (cap < 0) { const char* e = getenv("SERL_PCONV_DGW_CUS"); cap = e ? atoi(e) : 0; }
  const int cus = pconv_cus(ncu[dev]);
  int gx = (cap > 0 ? (cap < cus ? cap : cus) : cus * 3 / 4) / ngroups;
  if (gx < 1) gx = 1;
  if (gx > nitems) gx = nitems;
  gx = cdiv(nitems, cdiv(nitems, gx));
  hipLaunchKernelGGL(pconv_dgradw_kernel, dim3(gx, 1, ngroups), dim3(PDgradWCfg::NT), PDgradWCfg::LDS, st, a);
  SERL_CHECK_LAUNCH(name);
  return SERL_OK;
}

// ------------------------------------------------------------------------------------------------ conv1 + conv2 fused
// The first two layers of a NO-GRADIENT encoder pass (target critic, actor) in one kernel: uint8 frames -> conv 3x3/2
// (C1 <= 10 -> 32) + bias + ReLU -> conv 3x3/2 (32 -> 64) + bias + ReLU.  The first layer's output (508 KB per 128 x 128
// image in S format, the largest tensor of the encoder) is never written to HBM: per item (R = 4 output rows of one
// image) the block DMA-copies the 4R + 3 uint8 rows it needs (22 KB), computes the 2R + 1 rows of layer-1 output it
// needs on the matrix cores and writes them, rounded to S format exactly as serl_convs_fwd would store them, into the
// same LDS patch the stand-alone kernel fills by DMA; the second layer's K loop and output path are that kernel's.
// Same k order and MFMA sequence as the two im2col launches (uint8 layer: k' = ky * 32 + kx * C1 + c, products
// a_hi * b_lo + a_hi * b_hi; the bytes past a row's run meet zero weights): bit-identical to running them one after
// the other.  Layer-1 halo rows are computed twice (9/8 of the layer-1 work).
struct PConvFusedArgs {
  const unsigned char* u8[MAXG];      // [B][IH1][IW1][C1]
  const unsigned short* wf1[MAXG];    // S format [32][2][KH * 32]
  const float* bias1[MAXG];
  const unsigned short* wf2[MAXG];    // S format [64][2][288]
  const float* bias2[MAXG];
  unsigned short* y[MAXG];            // layer-2 output, S format [B][OH][OW][64]
  unsigned char* rmask[MAXG];
  // the pass whose backward follows also keeps layer 1's output (S format [B][IH][IW][32]) and its ReLU byte mask
  // ([B*IH*IW][4]): written from the LDS patch by the two-set kernel, nullptr for the no-gradient passes
  unsigned short* y1[MAXG];
  unsigned char* rmask1[MAXG];
  int B, IH1, IW1, C1, IH, IW, OH, OW, nbands, relu;     // IH, IW: layer-1 output = layer-2 input
  FastDiv d_bands;
};

struct PFusedCfg {
  typedef PFwdCfg<32, 64, 4, 32> G2;
  static constexpr int NT = 512, R = 4;
  static constexpr int U8ROWS = 4 * R + 3;               // uint8 rows per item
  static constexpr int RB1MAX = 1152;                    // bytes per uint8 row (IW1 * C1), multiple of 16
  static constexpr int U8BUF = 3 * NT * 16;              // 24 KB >= U8ROWS * RB1MAX + slack, 3 DMA pieces per thread
  static constexpr int W1IMG = 6 * 2 * 32 * 32;          // layer-1 weights as B fragments: [k16 step][plane][f][2 x 16 B]
  static constexpr int LDS = G2::PATCH + 2 * U8BUF + W1IMG;
  static_assert(U8ROWS * RB1MAX + 64 <= U8BUF && LDS <= 160 * 1024 && G2::ROWB == 16 * NT, "LDS");
};

__global__ __launch_bounds__(512, 1) void pconv_fused12_kernel(const PConvFusedArgs a) {
  typedef PFusedCfg::G2 G;
  constexpr int C = G::C, F = G::F, R = G::R, OWP = G::OWP, NT = G::NT;
  constexpr int REC = G::REC, ROWB = G::ROWB, PARB = OWP * REC;
  constexpr int KS = C / 16, NS = 9 * KS;
  constexpr int Kp = G::KP, WROW = G::WROW;
  constexpr int U8BUF = PFusedCfg::U8BUF;
  extern __shared__ __attribute__((aligned(16))) unsigned char p_smem[];
  unsigned char* const patch = p_smem;                       // layer-1 output of the item (S format, swizzled image)
  unsigned char* const u8buf = p_smem + G::PATCH;            // [2][U8BUF]

  const int z = blockIdx.z;
  const int nitems = a.B * a.nbands;
  const int per = (nitems + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it0 = blockIdx.x * per, it1 = min(nitems, it0 + per);
  if (it0 >= it1) return;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int wm = wid / G::NTL, wn = wid - wm * G::NTL;
  const int RB1 = a.IW1 * a.C1;

  // ---- layer-2 weights -> fragments in registers (as pconv_fwd_kernel), layer-1 weights straight from global memory
  bf16x8_t rbh[NS], rbl[NS];
  {
    constexpr int CPR = Kp / 8;
    constexpr int NWC = 2 * F * CPR, NWL = (NWC + NT - 1) / NT;
    const char* wsrc = reinterpret_cast<const char*>(a.wf2[z]);
    u32x4_t wv[NWL];
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int q = tid + i * NT;
      wv[i] = *reinterpret_cast<const u32x4_t*>(q < NWC ? wsrc + (size_t)q * 16 : reinterpret_cast<const char*>(g_zero_page));
    }
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int q = tid + i * NT;
      if (q < NWC) {
        const int row = q / CPR, kc = q - row * CPR;
        *reinterpret_cast<u32x4_t*>(p_smem + ((row & 1) * F + (row >> 1)) * WROW + kc * 16) = wv[i];
      }
    }
    __syncthreads();
    const unsigned char* wrow = p_smem + (wn * 32 + r) * WROW + h * 16;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      rbh[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(wrow + s * 32));
      rbl[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(wrow + F * WROW + s * 32));
    }
  }
  // layer-1 weights: fragment image in LDS (lane (r = f, h) reads k' = 16 s + 8 h .. + 7; the two halves of a row swap
  // on odd 8-row groups so that a 16-lane group of ds_read_b128 covers 16 distinct bank groups)
  unsigned char* const W1s = p_smem + G::PATCH + 2 * U8BUF;
  for (int id = tid; id < 6 * 2 * 32 * 2; id += NT) {
    const int hh = id & 1, f = (id >> 1) & 31, plane = (id >> 6) & 1, s6 = id >> 7;
    *reinterpret_cast<u32x4_t*>(W1s + ((s6 * 2 + plane) * 32 + f) * 32 + ((hh ^ ((f >> 3) & 1)) << 4)) =
        *reinterpret_cast<const u32x4_t*>(a.wf1[z] + (size_t)(f * 2 + plane) * 96 + s6 * 16 + hh * 8);
  }
  const unsigned char* const w1p = W1s + r * 32 + ((h ^ ((r >> 3) & 1)) << 4);
  const float b1 = a.bias1[z][r];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();            // weight image consumed: the LDS is free for the patches

  const int ox = r & (OWP - 1);
  const int arow = 2 * min(wm, R - 1) * ROWB;
  const int rec0 = ox * REC, rec1 = (ox + 1) * REC;
  const int sw0 = pconv_swz<REC>(ox) << 4, sw1 = pconv_swz<REC>(ox + 1) << 4;
  constexpr int LDC = F + 4;
  constexpr int CH = F / 8, RPP = NT / CH;
  const int n = (tid % CH) * 8;
  float bs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bs[j] = a.bias2[z][n + j];
  unsigned short* const out = a.y[z];
  unsigned char* const mk = a.rmask[z];

  // uint8 rows 4*oy0 .. 4*oy0 + 4*rows + 2 of image b: one contiguous span, copied linearly
  auto dma_u8 = [&](int item, unsigned char* dstbuf) {
    const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
    const int oy0 = band * R;
    const int nbytes = (4 * min(R, a.OH - oy0) + 3) * RB1;
    const char* src = reinterpret_cast<const char*>(a.u8[z]) + ((size_t)(b * a.IH1 + 4 * oy0)) * RB1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int off = ((i * 8 + wid) * 64 + lane) * 16;
      const char* g = off < nbytes ? src + off : reinterpret_cast<const char*>(g_zero_page);
      __builtin_amdgcn_global_load_lds((pconv_gptr_t)g, (pconv_lptr_t)(dstbuf + (i * 8 + wid) * 1024), 16, 0, 0);
    }
  };

  dma_u8(it0, u8buf);
  int pb = 0;
  for (int item = it0; item < it1; ++item, pb ^= 1) {
    const unsigned char* ub = u8buf + pb * U8BUF;
    const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
    const int oy0 = band * R;
    const int rows_here = min(R, a.OH - oy0);
    const int nrows1 = 2 * rows_here + 1;                    // layer-1 output rows of this item
    // uint8 rows landed; previous item's stores stay in flight (counted wait, see pconv_fwd_kernel); every wave is
    // done with the previous item's output tile, which lives in the patch
    static_assert(G::MT * 32 / RPP == 2, "store count of the counted wait");
    if (item == it0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (mk != nullptr) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // The last layer-1 row of a band is the first row of the next band of the same image: when this block computed it
    // for the previous item it is copied (8 KB, LDS to LDS) instead of recomputed -- 16 tiles = two full rounds of the
    // 8 waves instead of 18 = three.  (The output tile of the previous item overlays rows 0..4 only.)
    const bool reuse = item > it0 && band > 0;
    if (reuse && wid >= 6) {
      // copied by the very waves that overwrite that half of the last row in their second round (tiles 16, 17 belong
      // to waves 6, 7): program order inside a wave puts the read before the write, no barrier needed
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int off = (k >> 1) * PARB + (wid - 6) * 16 * REC + (k & 1) * 1024 + lane * 16;
        *reinterpret_cast<u32x4_t*>(patch + off) = *reinterpret_cast<const u32x4_t*>(patch + 2 * R * ROWB + off);
      }
    }
    if (item + 1 < it1) dma_u8(item + 1, u8buf + (pb ^ 1) * U8BUF);

    // ---- layer 1: 2 x nrows1 tiles of 32 pixels (row y1, columns 32 half .. + 31), round-robin over the 8 waves
#ifdef PF_NO_C1      // ablation builds
    if (item == it0)
#endif
    for (int t = (reuse ? 2 : 0) + wid; t < 2 * nrows1; t += 8) {
      const int y1 = t >> 1, ox1 = (t & 1) * 32 + r;
      f32x16 acc1;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc1[q] = 0.f;
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        // 8 consecutive bytes (kx, c) of input row 2 y1 + ky at pixel 2 ox1: 2-byte aligned, so three aligned dwords
        const uint32_t ad = (uint32_t)((2 * y1 + (s >> 1)) * RB1 + 2 * ox1 * a.C1 + (s & 1) * 16 + 8 * h);
        const uint32_t* p = reinterpret_cast<const uint32_t*>(ub + (ad & ~3u));
        const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
        const uint32_t lo4 = __builtin_amdgcn_alignbyte(d1, d0, ad & 2u);
        const uint32_t hi4 = __builtin_amdgcn_alignbyte(d2, d1, ad & 2u);
        const uint2 e0 = u8x4_to_bf16(lo4), e1 = u8x4_to_bf16(hi4);
        const u32x4_t av = {e0.x, e0.y, e1.x, e1.y};
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, av);
        const bf16x8_t w1h = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(w1p + (s * 2) * 1024));
        const bf16x8_t w1l = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(w1p + (s * 2 + 1) * 1024));
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, w1l, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, w1h, acc1, 0, 0, 0);
      }
      // bias + ReLU + S-format rounding, 2 bytes per (pixel, channel, plane) into the swizzled patch.  Pixel of
      // accumulator register q: ix = 32 half + 4 h + (q & 3) + 8 (q >> 2), so record j = ix >> 1 and its swizzle
      // (j >> 1) & 7 = h + 2 (q >> 2) split into a lane part and a compile-time part: one XOR per store.  Columns
      // past the image (ix = 63, or a narrower layer) go to records no valid lane reads.
      unsigned char* const tile = patch + y1 * 2 * PARB + ((t & 1) * 16 + 2 * h) * REC + (r & 7) * 2;
      const int ch0 = ((r >> 3) ^ h) << 4;
#pragma unroll
      for (int q = 0; q < 16; q += 2) {
        const float v0 = fmaxf(fmaf(acc1[q], 1.0f / 255.0f, b1), 0.f), v1 = fmaxf(fmaf(acc1[q + 1], 1.0f / 255.0f, b1), 0.f);
        uint32_t hh, ll;
        split_pair(v0, v1, hh, ll);
        // q even: parity 0, q + 1: parity 1, same j
        unsigned char* rec = tile + (((q & 3) >> 1) + 4 * (q >> 2)) * REC;
        const int co = ch0 ^ ((2 * (q >> 2)) << 4);
        *reinterpret_cast<unsigned short*>(rec + co) = (unsigned short)hh;
        *reinterpret_cast<unsigned short*>(rec + (co ^ 64)) = (unsigned short)ll;
        *reinterpret_cast<unsigned short*>(rec + PARB + co) = (unsigned short)(hh >> 16);
        *reinterpret_cast<unsigned short*>(rec + PARB + (co ^ 64)) = (unsigned short)(ll >> 16);
      }
    }
    pconv_lds_barrier();

    // ---- layer 2 (pconv_fwd_kernel's K loop and output path)
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#ifdef PF_NO_C2
    if (item == it0)
#endif
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int tapoff = arow + ky * ROWB + (kx & 1) * PARB + (kx == 2 ? rec1 : rec0);
      const int sw = kx == 2 ? sw1 : sw0;
#pragma unroll
      for (int kc = 0; kc < KS; ++kc) {
        const int chh = ((kc * 2 + h) << 4) ^ sw, chl = ((kc * 2 + h + C / 8) << 4) ^ sw;
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(patch + tapoff + chh));
        const bf16x8_t al = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(patch + tapoff + chl));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, rbh[tap * KS + kc], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, rbl[tap * KS + kc], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, rbh[tap * KS + kc], acc, 0, 0, 0);
      }
    }
    pconv_lds_barrier();

    float* const Cs = reinterpret_cast<float*>(patch);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      Cs[row * LDC + wn * 32 + r] = acc[q];
    }
    pconv_lds_barrier();
    const uint32_t cs_base = (uint32_t)(size_t)(pconv_lptr_t)Cs;
#pragma unroll
    for (int pass = 0; pass < G::MT * 32 / RPP; ++pass) {
      const int row0 = tid / CH + pass * RPP;
      const int oyl = min(row0 / OWP, rows_here - 1), oxx = min(row0 & (OWP - 1), a.OW - 1);
      const int row = oyl * OWP + oxx;
      const size_t pix = (size_t)(b * a.OH + oy0 + oyl) * a.OW + oxx;
      u32x4_t c0, c1;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(c0), "=&v"(c1) : "v"(cs_base + (uint32_t)(row * LDC + n) * 4u) : "memory");
      float v[8] = {__uint_as_float(c0[0]), __uint_as_float(c0[1]), __uint_as_float(c0[2]), __uint_as_float(c0[3]),
                    __uint_as_float(c1[0]), __uint_as_float(c1[1]), __uint_as_float(c1[2]), __uint_as_float(c1[3])};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = fmaf(v[j], 1.0f, bs[j]);
        if (a.relu) v[j] = fmaxf(v[j], 0.f);
      }
      u32x4_t hi, lo;
      uint32_t hh, ll;
      split_pair(v[0], v[1], hh, ll); hi[0] = hh; lo[0] = ll;
      split_pair(v[2], v[3], hh, ll); hi[1] = hh; lo[1] = ll;
      split_pair(v[4], v[5], hh, ll); hi[2] = hh; lo[2] = ll;
      split_pair(v[6], v[7], hh, ll); hi[3] = hh; lo[3] = ll;
      unsigned short* dst = out + pix * 2 * F + n;
      *reinterpret_cast<u32x4_t*>(dst) = hi;
      *reinterpret_cast<u32x4_t*>(dst + F) = lo;
      if (mk != nullptr) {
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bits |= (hi[j] & 0x7fffu) ? (1u << (2 * j)) : 0u;
          bits |= (hi[j] & 0x7fff0000u) ? (2u << (2 * j)) : 0u;
        }
        mk[pix * (F >> 3) + (n >> 3)] = (unsigned char)bits;
      }
    }
  }
}

static int pconv_fused12_launch(const PConvFusedArgs& a, int ngroups, hipStream_t st, const char* name) {
  static bool opted[64] = {};
  static int ncu[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), name);
  SERL_CHECK_ARG(dev >= 0 && dev < 64, "%s: device ordinal %d out of range", name, dev);
  if (!opted[dev]) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)pconv_fused12_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PFusedCfg::LDS), name);
    SERL_CHECK_HIP(hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev), name);
    opted[dev] = true;
  }
  const int nitems = a.B * a.nbands;
  int gx = pconv_cus(ncu[dev]) / ngroups;
  if (gx < 1) gx = 1;
  if (gx > nitems) gx = nitems;
  hipLaunchKernelGGL(pconv_fused12_kernel, dim3(gx, 1, ngroups), dim3(PFusedCfg::NT), PFusedCfg::LDS, st, a);
  SERL_CHECK_LAUNCH(name);
  return SERL_OK;
}

// ------------------------------------------------------------------------------------------------ conv1 + conv2, two wave sets
// pconv_fused12_kernel runs its phases one after the other on all 8 waves: the layer-1 phase is VALU work (uint8 ->
// bf16 conversion, S-format rounding, 2-byte LDS writes: ~240 VALU instructions per 12 MFMAs), the layer-2 phase is
// MFMAs only, the output phase VALU + stores again -- the matrix cores idle through two thirds of an item (0.27 of
// their roof).  Here the block is split into two sets of four waves (one wave of each set per SIMD) that work half a
// period apart on items of R = 2 output rows: while set s converts and multiplies layer 1 of item i + 1 into its own
// patch, set s^1 runs the layer-2 K loop and the output path of item i from its patch, so a SIMD always has one
// VALU-bound and one MFMA-bound wave to pick from.  One block barrier per slot; a set's patch, uint8 buffer and the
// output staging tile of a wave are private to it, the only cross-set access is the copy of the shared halo row (last
// layer-1 row of item i = first of item i + 1), read from the other set's patch while that set only reads it.
// Same operands, k order and MFMA sequence per output as the one-set kernel: bit-identical.
struct PFusedPPCfg {
  static constexpr int C = 32, F = 64, R = 2, OWP = 32, NT = 512;
  static constexpr int REC = 4 * C, PARB = OWP * REC, ROWB = 2 * PARB, NR = 2 * R + 1, PATCH = NR * ROWB;
  static constexpr int KP = 9 * C, WROW = KP * 2 + 16, WIMG = 2 * F * WROW;
  static constexpr int U8ROWS = 4 * R + 3;               // uint8 rows per item
  static constexpr int RB1MAX = 1152;
  static constexpr int U8BUF = 13 * 1024;                // 3 DMA pieces per lane of a set + one shared piece
  static constexpr int W1IMG = 6 * 2 * 32 * 32;
  static constexpr int LDT = 36, STG = 32 * LDT * 4;     // fp32 output tile of one wave
  static constexpr int LDS = 2 * PATCH + 2 * U8BUF + W1IMG + 8 * STG;
  static_assert(U8ROWS * RB1MAX + 64 <= U8BUF && WIMG <= 2 * PATCH && LDS <= 160 * 1024, "LDS");
  static_assert(RB1MAX == PFusedCfg::RB1MAX && R * 2 == PFusedCfg::R, "band geometry shared with pconv_fused12_kernel");
};

#ifdef PFPP_TRACE       // phase timestamps of block 0 (slots 8..11), tools/pf_trace.py
__device__ uint32_t pf_trace[256];
#define PF_TP(i) { if (k >= 8 && k < 12 && lane == 0) reinterpret_cast<uint32_t*>(p_smem + PFusedPPCfg::LDS)[((k - 8) * 8 + wid) * 8 + (i)] = (uint32_t)clock64(); }
#else
#define PF_TP(i)
#endif
#ifdef PFPP_TRACE
#define PFPP_TRACE_LDS 1024
#else
#define PFPP_TRACE_LDS 0
#endif
__global__ __launch_bounds__(512, 1) void pconv_fused12pp_kernel(const PConvFusedArgs a) {
  typedef PFusedPPCfg G;
  constexpr int C = G::C, F = G::F, R = G::R, NT = G::NT;
  constexpr int REC = G::REC, ROWB = G::ROWB, PARB = G::PARB;
  constexpr int KS = C / 16, NS = 9 * KS;
  constexpr int Kp = G::KP, WROW = G::WROW, LDT = G::LDT;
  extern __shared__ __attribute__((aligned(16))) unsigned char p_smem[];
  unsigned char* const patches = p_smem;                     // [2][PATCH]: layer-1 output, one per wave set
  unsigned char* const u8bufs = p_smem + 2 * G::PATCH;       // [2][U8BUF]
  unsigned char* const W1s = u8bufs + 2 * G::U8BUF;

  const int z = blockIdx.z;
  const int nitems = a.B * a.nbands;
  const int per = (nitems + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it0 = blockIdx.x * per, it1 = min(nitems, it0 + per);
  if (it0 >= it1) return;
  const int n_it = it1 - it0;
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
#ifdef PF12_SET_LOW
  const int set = wid & 1, w4 = wid >> 1;
#else
  const int set = wid >> 2, w4 = wid & 3;                    // waves w and w + 4 share a SIMD
#endif
  const int wm = w4 >> 1, wn = w4 & 1;
  const int RB1 = a.IW1 * a.C1;

  // ---- layer-2 weights -> fragments in registers, layer-1 weights -> fragment image in LDS (as pconv_fused12_kernel)
  bf16x8_t rbh[NS], rbl[NS];
  {
    constexpr int CPR = Kp / 8;
    constexpr int NWC = 2 * F * CPR, NWL = (NWC + NT - 1) / NT;
    const char* wsrc = reinterpret_cast<const char*>(a.wf2[z]);
    u32x4_t wv[NWL];
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int q = tid + i * NT;
      wv[i] = *reinterpret_cast<const u32x4_t*>(q < NWC ? wsrc + (size_t)q * 16 : reinterpret_cast<const char*>(g_zero_page));
    }
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int q = tid + i * NT;
      if (q < NWC) {
        const int row = q / CPR, kc = q - row * CPR;
        *reinterpret_cast<u32x4_t*>(p_smem + ((row & 1) * F + (row >> 1)) * WROW + kc * 16) = wv[i];
      }
    }
    __syncthreads();
    const unsigned char* wrow = p_smem + (wn * 32 + r) * WROW + h * 16;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      rbh[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(wrow + s * 32));
      rbl[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(wrow + F * WROW + s * 32));
    }
  }
  for (int id = tid; id < 6 * 2 * 32 * 2; id += NT) {
    const int hh = id & 1, f = (id >> 1) & 31, plane = (id >> 6) & 1, s6 = id >> 7;
    *reinterpret_cast<u32x4_t*>(W1s + ((s6 * 2 + plane) * 32 + f) * 32 + ((hh ^ ((f >> 3) & 1)) << 4)) =
        *reinterpret_cast<const u32x4_t*>(a.wf1[z] + (size_t)(f * 2 + plane) * 96 + s6 * 16 + hh * 8);
  }
  const unsigned char* const w1p = W1s + r * 32 + ((h ^ ((r >> 3) & 1)) << 4);
  const float b1 = a.bias1[z][r];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();            // weight image consumed: the LDS is free for the patches

  unsigned char* const patch = patches + set * G::PATCH;                   // this set's patch
  const unsigned char* const opatch = patches + (set ^ 1) * G::PATCH;      // the other set's (halo row copy only)
  unsigned char* const ubuf = u8bufs + set * G::U8BUF;
  float* const Cs = reinterpret_cast<float*>(W1s + G::W1IMG + wid * G::STG);
  const uint32_t cs_base = (uint32_t)(size_t)(pconv_lptr_t)Cs;
  const uint32_t patch_l = (uint32_t)(size_t)(pconv_lptr_t)patch, ub_l = (uint32_t)(size_t)(pconv_lptr_t)ubuf;
  const uint32_t w1_l = (uint32_t)(size_t)(pconv_lptr_t)w1p;
  const int rec0 = r * REC, rec1 = (r + 1) * REC;
  const int sw0 = pconv_swz<REC>(r) << 4, sw1 = pconv_swz<REC>(r + 1) << 4;
  const int n8 = wn * 32 + (lane & 3) * 8;
  float bs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) bs[j] = a.bias2[z][n8 + j];
  unsigned short* const out = a.y[z];
  unsigned char* const mk = a.rmask[z];
  unsigned short* const y1g = a.y1[z];
  unsigned char* const mk1 = a.rmask1[z];

  // uint8 rows 4*oy0 .. 4*oy0 + 4*rows + 2 of image b -> this set's buffer: 3 pieces per lane + a 13th KB that all four
  // waves copy (the same bytes to the same place)
  auto dma_u8 = [&](int item) {
    const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
    const int oy0 = band * R;
    const int nbytes = (4 * min(R, a.OH - oy0) + 3) * RB1;
    const char* src = reinterpret_cast<const char*>(a.u8[z]) + ((size_t)(b * a.IH1 + 4 * oy0)) * RB1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pc = i < 3 ? i * 4 + w4 : 12;
      const int off = (pc * 64 + lane) * 16;
#ifdef PFPP_ZERO_SRC
      const char* g = reinterpret_cast<const char*>(g_zero_page) + (off < nbytes ? (off & 4080) : 0);
#else
      const char* g = off < nbytes ? src + off : reinterpret_cast<const char*>(g_zero_page);
#endif
      __builtin_amdgcn_global_load_lds((pconv_gptr_t)g, (pconv_lptr_t)(ubuf + pc * 1024), 16, 0, 0);
    }
  };

  dma_u8(min(it0 + set, it1 - 1));         // set s owns items it0 + s, it0 + s + 2, ...
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // slot k: the set with (k & 1) == set computes layer 1 of item it0 + k, the other one layer 2 of item it0 + k - 1
  for (int k = 0; k <= n_it; ++k) {
    PF_TP(0)
    if ((k & 1) == set) {
      if (k < n_it) {
        const int item = it0 + k;
        const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
        const int nrows1 = 2 * min(R, a.OH - band * R) + 1;     // layer-1 output rows of this item
        // the first layer-1 row is the last one of the previous item of this block (other set's patch, written one
        // slot ago, only read by its owner now): copied, 8 tiles = two rounds of the four waves instead of 10 = three
        const bool reuse = k > 0 && band > 0;
        if (reuse) {
#pragma unroll
          for (int kk = 0; kk < 2; ++kk) {
            const int off = (w4 * 2 + kk) * 1024 + lane * 16;
            *reinterpret_cast<u32x4_t*>(patch + off) = *reinterpret_cast<const u32x4_t*>(opatch + 2 * R * ROWB + off);
          }
        }
        // Four tiles of 32 pixels per row pair, two (or three) per wave.  One wave per SIMD works here, so nobody else
        // hides the LDS latency: the uint8 dwords run two k16 steps ahead and the weight fragments one step ahead of the
        // MFMAs (register rings; hipcc's own order was read -> wait -> convert -> read -> wait -> two dependent MFMAs,
        // 3500 cycles per tile), across the tile boundary too.
        PF_TP(1)
        int t = (reuse ? 2 : 0) + w4;
        const int tend = 2 * nrows1;
#ifdef PFPP_NO_L1      // ablation builds
        if (k > 1) t = tend;
#endif
        if (t < tend) {
          // 8 consecutive bytes (kx, c) of input row 2 y1 + ky at pixel 2 ox1: 2-byte aligned, so three aligned dwords
          auto tile_base = [&](int tt) { return (uint32_t)((tt >> 1) * 2 * RB1 + 2 * ((tt & 1) * 32 + r) * a.C1 + 8 * h); };
          u32x2_t D01[3];
          uint32_t D2[3];
          u32x4_t Wh[2], Wl[2];
          uint32_t base = tile_base(t);
          uint32_t ab = ub_l + (base & ~3u);
          // software pipeline: the uint8 dwords of step s + 3 and the weight fragments of step s + 1 are requested in
          // step s, and the A fragment of step s + 1 is converted BETWEEN the two (dependent) MFMAs of step s, whose
          // 32-cycle shadows would otherwise be stalls of this wave
          PF_LD_D(D01[0], D2[0], ab, 0);
          PF_LD_D(D01[1], D2[1], ab, 16);
          PF_LD_D(D01[2], D2[2], ab + RB1, 0);
          PF_LD_2X128(Wh[0], Wl[0], w1_l, w1_l, 0, 1024);
          asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(D01[0]), "+v"(D2[0]) : : "memory");
          u32x4_t acur;
          {
            const uint32_t lo4 = __builtin_amdgcn_alignbyte(D01[0][1], D01[0][0], base & 2u);
            const uint32_t hi4 = __builtin_amdgcn_alignbyte(D2[0], D01[0][1], base & 2u);
            const uint2 e0 = u8x4_to_bf16(lo4), e1 = u8x4_to_bf16(hi4);
            acur = u32x4_t{e0.x, e0.y, e1.x, e1.y};
          }
          for (; t < tend; t += 4) {
            const uint32_t sh = base & 2u;                       // (rows are multiples of 16 bytes: the same for all steps)
            const uint32_t nbase = tile_base(t + 4 < tend ? t + 4 : t);
            const uint32_t nsh = nbase & 2u;
            const uint32_t nab = ub_l + (nbase & ~3u);           // the next tile's first steps (the last tile: its own again)
            const uint32_t ab1 = ab + RB1, ab2 = ab1 + RB1, nab1 = nab + RB1;
            f32x16 acc1;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc1[q] = 0.f;
#define PF_L1_STEP(s, ADDR, IMM, SH)                                                                                    \
            {                                                                                                           \
              PF_LD_D(D01[(s) % 3], D2[(s) % 3], ADDR, IMM);                                                            \
              PF_LD_2X128(Wh[((s) + 1) & 1], Wl[((s) + 1) & 1], w1_l, w1_l, (((s) + 1) % 6) * 2048, (((s) + 1) % 6) * 2048 + 1024); \
              asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(D01[((s) + 1) % 3]), "+v"(D2[((s) + 1) % 3]), "+v"(Wh[(s) & 1]), "+v"(Wl[(s) & 1]) : : "memory"); \
              const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, acur);                                                   \
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8_t, Wl[(s) & 1]), acc1, 0, 0, 0); \
              __builtin_amdgcn_sched_barrier(0);                                                                        \
              const uint32_t lo4 = __builtin_amdgcn_alignbyte(D01[((s) + 1) % 3][1], D01[((s) + 1) % 3][0], SH);        \
              const uint32_t hi4 = __builtin_amdgcn_alignbyte(D2[((s) + 1) % 3], D01[((s) + 1) % 3][1], SH);            \
              const uint2 e0 = u8x4_to_bf16(lo4);                                                                       \
              __builtin_amdgcn_sched_barrier(0);                                                                        \
              acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8_t, Wh[(s) & 1]), acc1, 0, 0, 0); \
              __builtin_amdgcn_sched_barrier(0);                                                                        \
              const uint2 e1 = u8x4_to_bf16(hi4);                                                                       \
              acur = u32x4_t{e0.x, e0.y, e1.x, e1.y};                                                                   \
              __builtin_amdgcn_sched_barrier(0);                                                                        \
            }
            PF_L1_STEP(0, ab1, 16, sh) PF_L1_STEP(1, ab2, 0, sh) PF_L1_STEP(2, ab2, 16, sh)
            PF_L1_STEP(3, nab, 0, sh) PF_L1_STEP(4, nab, 16, sh) PF_L1_STEP(5, nab1, 0, nsh)
#undef PF_L1_STEP
#ifdef PFPP_TRACE
            if (t < 6) PF_TP(3)
#endif
            // bias + ReLU + S-format rounding, 2 bytes per (pixel, channel, plane) into the swizzled patch (layout and
            // swizzle arithmetic: pconv_fused12_kernel).  (Weights as the A operand would leave a lane with four
            // consecutive channels of one pixel, i.e. 8 ds_write_b64 instead of 32 ds_write_b16 per tile, and is still
            // bit-identical -- but needs the 16 biases of those channels in registers, which spills; measured no faster.)
            const int y1 = t >> 1;
            unsigned char* const tile = patch + y1 * 2 * PARB + ((t & 1) * 16 + 2 * h) * REC + (r & 7) * 2;
            const int ch0 = ((r >> 3) ^ h) << 4;
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
              const float v0 = fmaxf(fmaf(acc1[q], 1.0f / 255.0f, b1), 0.f), v1 = fmaxf(fmaf(acc1[q + 1], 1.0f / 255.0f, b1), 0.f);
              uint32_t hh, ll;
              split_pair(v0, v1, hh, ll);
              unsigned char* rec = tile + (((q & 3) >> 1) + 4 * (q >> 2)) * REC;
              const int co = ch0 ^ ((2 * (q >> 2)) << 4);
              *reinterpret_cast<unsigned short*>(rec + co) = (unsigned short)hh;
              *reinterpret_cast<unsigned short*>(rec + (co ^ 64)) = (unsigned short)ll;
              *reinterpret_cast<unsigned short*>(rec + PARB + co) = (unsigned short)(hh >> 16);
              *reinterpret_cast<unsigned short*>(rec + PARB + (co ^ 64)) = (unsigned short)(ll >> 16);
            }
#ifdef PFPP_TRACE
            if (t < 6) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); PF_TP(4) }
#endif
            base = nbase; ab = nab;
          }
          // the reads issued for a tile that does not come must land before their registers are anyone else's
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(D01[0]), "+v"(D2[0]), "+v"(D01[1]), "+v"(D2[1]), "+v"(D01[2]), "+v"(D2[2]),
                       "+v"(Wh[0]), "+v"(Wl[0]), "+v"(Wh[1]), "+v"(Wl[1]) : : "memory");
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PF_TP(2)
    } else if (k >= 1) {
      const int item = it0 + k - 1;
      const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
      const int oy0 = band * R;
      const int oyl = min(wm, min(R, a.OH - oy0) - 1);       // a band of one row: both row waves compute and store it
      // this set's next item; past the end a copy nobody reads (branch-free issue, same store count for the wait)
      dma_u8(min(it0 + k + 1, it1 - 1));
      PF_TP(1)
      const int arow = 2 * oyl * ROWB;
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      // A fragments two k16 steps ahead of the MFMAs (ring of three register pairs, read by hand as above)
      {
        u32x4_t Ah[3], Al[3];
        uint32_t pa[2][KS][2];               // [kx == 2][kc][plane]: lane address of the fragment, tap offsets are immediates
#pragma unroll
        for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
          for (int kc = 0; kc < KS; ++kc)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
              pa[x2][kc][pl] = patch_l + (uint32_t)(arow + (x2 ? rec1 : rec0) + (((kc * 2 + h + pl * (C / 8)) << 4) ^ (x2 ? sw1 : sw0)));
#define PF_LD_A(slot, s)                                                                                                \
        PF_LD_2X128(Ah[slot], Al[slot], pa[((s) / KS) % 3 == 2][(s) % KS][0], pa[((s) / KS) % 3 == 2][(s) % KS][1],     \
                    ((s) / KS / 3) * ROWB + ((((s) / KS) % 3) & 1) * PARB, ((s) / KS / 3) * ROWB + ((((s) / KS) % 3) & 1) * PARB)
#define PF_L2_STEP(s, CNT)                                                                                              \
        {                                                                                                               \
          if ((s) + 2 < NS) PF_LD_A(((s) + 2) % 3, ((s) + 2) % NS);                                                     \
          asm volatile("s_waitcnt lgkmcnt(" #CNT ")" : "+v"(Ah[(s) % 3]), "+v"(Al[(s) % 3]) : : "memory");              \
          const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, Ah[(s) % 3]), al = __builtin_bit_cast(bf16x8_t, Al[(s) % 3]); \
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, rbh[s], acc, 0, 0, 0);                                      \
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, rbl[s], acc, 0, 0, 0);                                      \
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, rbh[s], acc, 0, 0, 0);                                      \
          __builtin_amdgcn_sched_barrier(0);                                                                            \
        }
        static_assert(NS == 18 && KS == 2, "unrolled by hand");
#ifdef PFPP_NO_L2
        if (k <= 2)
#endif
        {
        PF_LD_A(0, 0); PF_LD_A(1, 1);
        PF_L2_STEP(0, 4) PF_L2_STEP(1, 4) PF_L2_STEP(2, 4) PF_L2_STEP(3, 4) PF_L2_STEP(4, 4) PF_L2_STEP(5, 4)
        PF_L2_STEP(6, 4) PF_L2_STEP(7, 4) PF_L2_STEP(8, 4) PF_L2_STEP(9, 4) PF_L2_STEP(10, 4) PF_L2_STEP(11, 4)
        PF_L2_STEP(12, 4) PF_L2_STEP(13, 4) PF_L2_STEP(14, 4) PF_L2_STEP(15, 4) PF_L2_STEP(16, 2) PF_L2_STEP(17, 0)
        }
#undef PF_L2_STEP
#undef PF_LD_A
      }
      PF_TP(2)
      // ---- output through the wave's own fp32 tile (32 pixels x 32 channels): 8 channels of one pixel per lane
#pragma unroll
      for (int q = 0; q < 16; ++q) Cs[((q & 3) + 8 * (q >> 2) + 4 * h) * LDT + r] = acc[q];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int oxx = min((lane >> 2) + 16 * pass, a.OW - 1);
        const size_t pix = (size_t)(b * a.OH + oy0 + oyl) * a.OW + oxx;
        u32x4_t c0, c1;
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(c0), "=&v"(c1) : "v"(cs_base + (uint32_t)(oxx * LDT + (lane & 3) * 8) * 4u) : "memory");
        float v[8] = {__uint_as_float(c0[0]), __uint_as_float(c0[1]), __uint_as_float(c0[2]), __uint_as_float(c0[3]),
                      __uint_as_float(c1[0]), __uint_as_float(c1[1]), __uint_as_float(c1[2]), __uint_as_float(c1[3])};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v[j] = fmaf(v[j], 1.0f, bs[j]);
          if (a.relu) v[j] = fmaxf(v[j], 0.f);
        }
        u32x4_t hi, lo;
        uint32_t hh, ll;
        split_pair(v[0], v[1], hh, ll); hi[0] = hh; lo[0] = ll;
        split_pair(v[2], v[3], hh, ll); hi[1] = hh; lo[1] = ll;
        split_pair(v[4], v[5], hh, ll); hi[2] = hh; lo[2] = ll;
        split_pair(v[6], v[7], hh, ll); hi[3] = hh; lo[3] = ll;
#ifdef PFPP_NO_STORE
        unsigned short* dst = out + (k > 2 ? 0 : pix * 2 * F) + n8 + (size_t)blockIdx.x * 4096;
#else
        unsigned short* dst = out + pix * 2 * F + n8;
#endif
        *reinterpret_cast<u32x4_t*>(dst) = hi;
        *reinterpret_cast<u32x4_t*>(dst + F) = lo;
        if (mk != nullptr) {
          uint32_t bits = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            bits |= (hi[j] & 0x7fffu) ? (1u << (2 * j)) : 0u;
            bits |= (hi[j] & 0x7fff0000u) ? (2u << (2 * j)) : 0u;
          }
          mk[pix * (F >> 3) + (n8 >> 3)] = (unsigned char)bits;
        }
      }
      PF_TP(3)
      // the next item's uint8 rows have landed (the DMA is older than this slot's 4 or 6 stores)
      if (mk != nullptr) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      PF_TP(4)
      // ---- the pass with a backward keeps layer 1's output: this item's patch rows go to memory as they are (the S-format
      // records of serl_convs_fwd(in_u8 = 1), unswizzled) with their ReLU byte mask.  A band owns its first 2 rows' worth
      // (the last row is the next band's first), the last band of an image its last row too.  Issued behind the counted
      // wait above: the stores stay in flight into the next slots (older than this set's next DMA).  Row y by wave y & 3;
      // 8 lanes per record (128 bytes contiguous), 16-lane groups read records j, j + 1 of one parity: distinct banks.
      if (y1g != nullptr) {
        const int nsave = 2 * min(R, a.OH - oy0) + (band == a.nbands - 1 ? 1 : 0);
        for (int y = w4; y < nsave; y += 4) {
          const size_t rowpix = ((size_t)b * a.IH + 2 * oy0 + y) * a.IW;
          u32x4_t v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int par = i & 1, j = (i >> 1) * 8 + (lane >> 3), c = lane & 7;
            const uint32_t la = patch_l + (uint32_t)(y * ROWB + par * PARB + j * REC + ((c ^ pconv_swz<REC>(j)) << 4));
            asm volatile("ds_read_b128 %0, %1" : "=v"(v[i]) : "v"(la) : "memory");
          }
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : : "memory");
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int par = i & 1, j = (i >> 1) * 8 + (lane >> 3), c = lane & 7;
            const int ix = 2 * j + par;
            if (ix < a.IW) {
              *reinterpret_cast<u32x4_t*>(y1g + (rowpix + ix) * 2 * C + c * 8) = v[i];
              if (mk1 != nullptr && c < 4) {
                uint32_t bits = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  bits |= (v[i][e] & 0x7fffu) ? (1u << (2 * e)) : 0u;
                  bits |= (v[i][e] & 0x7fff0000u) ? (2u << (2 * e)) : 0u;
                }
                mk1[(rowpix + ix) * (C >> 3) + c] = (unsigned char)bits;
              }
            }
          }
        }
      }
    }
    __builtin_amdgcn_s_barrier();
    PF_TP(5)
  }
#ifdef PFPP_TRACE
  if (blockIdx.x == 0 && z == 0) {
    __syncthreads();
    if (tid < 256) pf_trace[tid] = reinterpret_cast<uint32_t*>(p_smem + PFusedPPCfg::LDS)[tid];
  }
#endif
}

static int pconv_fused12pp_launch(const PConvFusedArgs& a, int ngroups, hipStream_t st, const char* name) {
  static bool opted[64] = {};
  static int ncu[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), name);
  SERL_CHECK_ARG(dev >= 0 && dev < 64, "%s: device ordinal %d out of range", name, dev);
  if (!opted[dev]) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)pconv_fused12pp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PFusedPPCfg::LDS + PFPP_TRACE_LDS), name);
    SERL_CHECK_HIP(hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev), name);
    opted[dev] = true;
  }
  const int nitems = a.B * a.nbands;
  int gx = pconv_cus(ncu[dev]) / ngroups;
  if (gx < 1) gx = 1;
  if (gx > nitems) gx = nitems;
  hipLaunchKernelGGL(pconv_fused12pp_kernel, dim3(gx, 1, ngroups), dim3(PFusedPPCfg::NT), PFusedPPCfg::LDS + PFPP_TRACE_LDS, st, a);
  SERL_CHECK_LAUNCH(name);
  return SERL_OK;
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[ky][kx][c][f] = sum over (image, oy, ox) x[2 oy + ky][2 ox + kx][c] * dz[oy][ox][f], dbias[f] = sum dz, for the
// 3x3 stride-2 layer with C = 32, F = 64.  The im2col kernel gathers its A operand (8 channels of one tap of one
// pixel per 16-byte load, every lane its own cache line) and re-reads dz once per 128-row tile of the 289 rows:
// 0.28 of the HBM roof.  Here an item is a band of R = 2 output rows of one image: 5 rows of x and 2 rows of dz are
// staged in LDS by LDS-DMA (double-buffered, layouts of pconv_fwd_kernel / pconv_dgrad_kernel), and both MFMA
// operands -- k = pixel runs along the records -- are read with ds_read_b64_tr_b16 (hardware transpose).  The
// 9 taps x 2 halves of F plus an all-ones row for the bias are 20 accumulator tiles = 2 per wave of a 10-wave block;
// they stay in registers over ALL items of the persistent block, which then writes one fp32 slab; the slabs are summed
// in a fixed order by wgrad_reduce_kernel (conv_geom.h) as before: reproducible, no atomics.
// four transposing LDS reads, issued and NOT waited for (the caller's s_waitcnt ties the outputs)
#define PCONV_TR4(o0, o1, o2, o3, a0, a1, a2, a3)                                                                      \
  asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %5\n\tds_read_b64_tr_b16 %2, %6\n\tds_read_b64_tr_b16 %3, %7" \
               : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)                                                             \
               : "v"((uint32_t)(a0)), "v"((uint32_t)(a1)), "v"((uint32_t)(a2)), "v"((uint32_t)(a3))                     \
               : "memory")

struct PConvWgradArgs {
  const unsigned short* x[MAXG];
  const unsigned short* dz[MAXG];
  float* slab;                        // [group][gridDim.x][289 * 64]
  int B, IH, IW, OH, OW, nbands;
  FastDiv d_bands;
};

struct PWgradCfg {
  static constexpr int C = 32, F = 64, R = 2, NW = 10, NT = 64 * NW;
  static constexpr int XREC = 4 * C, XPARB = 32 * XREC, XROWB = 2 * XPARB, XROWS = 2 * R + 1;
  static constexpr int XPATCH = XROWS * XROWB;              // 40 KB
  static constexpr int ZREC = 4 * F, ZROWB = 32 * ZREC;
  static constexpr int ZPATCH = R * ZROWB;                  // 16 KB
  static constexpr int BUF = XPATCH + ZPATCH;
  static constexpr int LDS = 2 * BUF;
  static constexpr int XPIECES = XPATCH / 16 / NT;          // 4
  static constexpr int ZSLOTS = ZPATCH / 16;                // 1024
  static constexpr int SLAB = (9 * C + 1) * F;              // floats per slab
  static_assert(XPATCH % (16 * NT) == 0 && LDS <= 160 * 1024, "LDS");
};

__global__ __launch_bounds__(640, 1) void pconv_wgrad_kernel(const PConvWgradArgs a) {
  typedef PWgradCfg G;
  constexpr int C = G::C, F = G::F, R = G::R, NW = G::NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char p_smem[];
  const int z = blockIdx.z;
  const int nitems = a.B * a.nbands;
  const int per = (nitems + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it0 = blockIdx.x * per, it1 = min(nitems, it0 + per);    // the launcher sizes the grid so that it0 < it1
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int nt = wid & 1, mt0 = (wid >> 1) * 2;              // N tile; M tiles mt0, mt0 + 1 (tap index, 9 = ones row)
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_cb = lane & 16;

  auto dma = [&](int item, unsigned char* buf) {
    const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
    const int oy0 = band * R;
    const char* xsrc = reinterpret_cast<const char*>(a.x[z]) + ((size_t)(b * a.IH + 2 * oy0) * a.IW) * G::XREC;
    const int xrows = min(G::XROWS, a.IH - 2 * oy0);
#pragma unroll 1
    for (int i = 0; i < G::XPIECES; ++i) {
      const int l = (i * NW + wid) * 64 + lane;
      const int rec = l >> 3, chp = l & 7;
      const int rr = rec >> 6, par = (rec >> 5) & 1, j = rec & 31;
      const int ix = 2 * j + par;
      const bool ok = rr < xrows && ix < a.IW;
      const char* g = ok ? xsrc + (size_t)(((rr * a.IW + ix) << 3) + (chp ^ pconv_swz<G::XREC>(j))) * 16
                         : reinterpret_cast<const char*>(g_zero_page);
      __builtin_amdgcn_global_load_lds((pconv_gptr_t)g, (pconv_lptr_t)(buf + (i * NW + wid) * 1024), 16, 0, 0);
    }
    const char* zsrc = reinterpret_cast<const char*>(a.dz[z]) + ((size_t)(b * a.OH + oy0) * a.OW) * G::ZREC;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if ((i * NW + wid) * 64 < G::ZSLOTS) {               // wave-uniform
        const int l = (i * NW + wid) * 64 + lane;
        const int rec = l >> 4, chp = l & 15;
        const int row = rec >> 5, ox = rec & 31;
        const bool ok = oy0 + row < a.OH && ox < a.OW;
        const char* g = ok ? zsrc + (size_t)(((row * a.OW + ox) << 4) + (chp ^ (ox & 15))) * 16
                           : reinterpret_cast<const char*>(g_zero_page);
        __builtin_amdgcn_global_load_lds((pconv_gptr_t)g, (pconv_lptr_t)(buf + G::XPATCH + (i * NW + wid) * 1024), 16, 0, 0);
      }
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
  // A fragment of the all-ones row: row 0 = 1.0 for every k
  const uint32_t one2 = r == 0 ? 0x3F803F80u : 0u;
  const u32x4_t onesv = {one2, one2, one2, one2};
  const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, onesv);

  const uint32_t smem_base = (uint32_t)(size_t)(pconv_lptr_t)p_smem;
  dma(it0, p_smem);
  int pb = 0;
  for (int item = it0; item < it1; ++item, pb ^= 1) {
    const uint32_t xp_base = smem_base + pb * G::BUF, zp_base = xp_base + G::XPATCH;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // this item's patches landed; every wave is done with the other buffer
    if (item + 1 < it1) dma(item + 1, p_smem + (pb ^ 1) * G::BUF);
#pragma unroll
    for (int st = 0; st < 2 * R; ++st) {       // k16 step: output row oyl, pixels ox = 16 sx + 0..15
      const int oyl = st >> 1, sx = st & 1;
      // this lane feeds the transposed reads with k-rows (pixels) ox0 and ox0 + 4 and 4 consecutive rows (channels).
      // The reads are hand-written: hipcc puts vmcnt(0) in front of the ds_read_tr builtin while a DMA is in flight.
      const int ox0 = 16 * sx + 8 * h + tr_q;
      s16x4_t bq[4], aq[2][4];
      {
        const int cofs = (32 * nt + tr_cb + 4 * tr_p) * 2;           // byte offset inside the hi plane of a dz record
        const int chunk = cofs >> 4, within = cofs & 15;
        const uint32_t r0 = zp_base + oyl * G::ZROWB + ox0 * G::ZREC + within, r1 = r0 + 4 * G::ZREC;
        const int s0 = ox0 & 15, s1 = (ox0 + 4) & 15;
        PCONV_TR4(bq[0], bq[1], bq[2], bq[3], r0 + ((chunk ^ s0) << 4), r1 + ((chunk ^ s1) << 4),
                  r0 + (((chunk + 8) ^ s0) << 4), r1 + (((chunk + 8) ^ s1) << 4));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int mt = mt0 + i;
        if (mt < 9) {                          // wave-uniform
          const int ky = mt / 3, kx = mt - 3 * ky;
          const int cofs = (tr_cb + 4 * tr_p) * 2;                    // byte offset inside the hi plane of an x record
          const int chunk = cofs >> 4, within = cofs & 15;
          const int j0 = ox0 + (kx >> 1), j1 = j0 + 4;
          const uint32_t row = xp_base + ((2 * oyl + ky) * 2 + (kx & 1)) * G::XPARB + within;
          const uint32_t r0 = row + j0 * G::XREC, r1 = row + j1 * G::XREC;
          const int s0 = pconv_swz<G::XREC>(j0), s1 = pconv_swz<G::XREC>(j1);
          PCONV_TR4(aq[i][0], aq[i][1], aq[i][2], aq[i][3], r0 + ((chunk ^ s0) << 4), r1 + ((chunk ^ s1) << 4),
                    r0 + (((chunk + 4) ^ s0) << 4), r1 + (((chunk + 4) ^ s1) << 4));
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) aq[i][e] = s16x4_t{0, 0, 0, 0};
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(aq[0][0]), "+v"(aq[0][1]), "+v"(aq[0][2]),
                     "+v"(aq[0][3]), "+v"(aq[1][0]), "+v"(aq[1][1]), "+v"(aq[1][2]), "+v"(aq[1][3]));
      typedef short s16x8_t __attribute__((ext_vector_type(8)));
      const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, (s16x8_t){bq[0][0], bq[0][1], bq[0][2], bq[0][3], bq[1][0], bq[1][1], bq[1][2], bq[1][3]});
      const bf16x8_t bl = __builtin_bit_cast(bf16x8_t, (s16x8_t){bq[2][0], bq[2][1], bq[2][2], bq[2][3], bq[3][0], bq[3][1], bq[3][2], bq[3][3]});
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (mt0 + i == 9) {                    // the bias row
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bl, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, bh, acc[i], 0, 0, 0);
        } else {
          const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, (s16x8_t){aq[i][0][0], aq[i][0][1], aq[i][0][2], aq[i][0][3], aq[i][1][0], aq[i][1][1], aq[i][1][2], aq[i][1][3]});
          const bf16x8_t al = __builtin_bit_cast(bf16x8_t, (s16x8_t){aq[i][2][0], aq[i][2][1], aq[i][2][2], aq[i][2][3], aq[i][3][0], aq[i][3][1], aq[i][3][2], aq[i][3][3]});
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
        }
      }
    }
  }
  // ---- one slab per block: rows (tap, c) in HWIO order, then the bias row
  float* slab = a.slab + ((size_t)z * gridDim.x + blockIdx.x) * G::SLAB;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int mt = mt0 + i;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int m = (q & 3) + 8 * (q >> 2) + 4 * h;
      if (mt < 9) slab[(size_t)(mt * C + m) * F + nt * 32 + r] = acc[i][q];
      else if (m == 0) slab[(size_t)(9 * C) * F + nt * 32 + r] = acc[i][q];
    }
  }
}

// returns the number of slabs per group through *nslab (0: the workspace cannot hold the slabs).  The slab count fixes the
// summation order, so it depends on the geometry and the device only -- the CU count, the split count `max_slabs` of
// wgrad_plan (what serl_conv2d_wgrad_workspace_bytes sizes a workspace for) and the item count -- never on how large the
// caller's workspace happens to be or on serl_convs_set_cu_limit: serl_convs_wgrad and serl_convs_wgrad_noalloc give the
// same bits.  workspace_bytes is a sufficiency check only.
static int pconv_wgrad_launch(PConvWgradArgs& a, int ngroups, int max_slabs, long long workspace_bytes, int* nslab,
                              hipStream_t st, const char* name) {
  static bool opted[64] = {};
  static int ncu[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), name);
  SERL_CHECK_ARG(dev >= 0 && dev < 64, "%s: device ordinal %d out of range", name, dev);
  if (!opted[dev]) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)pconv_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PWgradCfg::LDS), name);
    SERL_CHECK_HIP(hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev), name);
    opted[dev] = true;
  }
  const int nitems = a.B * a.nbands;
  static int cap = -1;                    // SERL_PCONV_WG_CUS: CUs this kernel spreads over (experiments with co-running streams)
  if (cap < 0) { const char* e = getenv("SERL_PCONV_WG_CUS"); cap = e ? atoi(e) : 0; }
  const int cus = ncu[dev] > 0 ? ncu[dev] : 256;
  int gx = (cap > 0 && cap < cus ? cap : cus) / ngroups;
  if (gx > max_slabs) gx = max_slabs;
  if (gx > nitems) gx = nitems;
  if (gx < 1) { *nslab = 0; return SERL_OK; }
  const int per = cdiv(nitems, gx);
  gx = cdiv(nitems, per);                    // every block has at least one item
  if ((long long)ngroups * gx * PWgradCfg::SLAB * 4 > workspace_bytes) { *nslab = 0; return SERL_OK; }
  *nslab = gx;
  hipLaunchKernelGGL(pconv_wgrad_kernel, dim3(gx, 1, ngroups), dim3(PWgradCfg::NT), PWgradCfg::LDS, st, a);
  SERL_CHECK_LAUNCH(name);
  return SERL_OK;
}

// ------------------------------------------------------------------------------------------------ weight gradient, wide layer
// The same weight gradient for C = 64, F = 128 (third layer of the encoder: x 31 x 31 x 64, dz 15 x 15 x 128).  The im2col
// kernel needs 103 split-K slabs per camera to fill the chip (61 MB written and read back per step, and the reduction
// behind it takes as long as the GEMM inside the update).  Here the 9 taps x 2 channel halves + the all-ones row are 19 M
// tiles, F is 4 N tiles: 76 accumulator tiles = 5 x 2 per wave of an 8-wave block (160 registers), persistent over all items
// of the block, which writes ONE slab.  An item is a band of R = 2 output rows of one image: 5 rows of x (256-byte records,
// [row][parity][j], 16 per parity) and 2 rows of dz (512-byte records, 16 per row: record 15 is zero) by LDS-DMA, x three
// and dz two buffers deep; one k16 step per output row, both operands by ds_read_b64_tr_b16 as in pconv_wgrad_kernel.
struct PWgradWCfg {
  static constexpr int C = 64, F = 128, R = 2, NW = 8, NT = 64 * NW;
  static constexpr int XREC = 4 * C, XPARB = 16 * XREC, XROWB = 2 * XPARB, XROWS = 2 * R + 1;
  static constexpr int XPATCH = XROWS * XROWB;              // 40 KB
  static constexpr int ZREC = 4 * F, ZROWB = 16 * ZREC;
  static constexpr int ZPATCH = R * ZROWB;                  // 16 KB
  // x bands three deep, dz bands two deep: per item the block computes for ~1 us, an LDS-DMA burst of a band takes 2-3 us to
  // land, so two x bands (the larger operand) are in flight while a third is multiplied
  static constexpr int NXB = 3, NZB = 2, OFF_Z = NXB * XPATCH, OFF_ONES = OFF_Z + NZB * ZPATCH;
  static constexpr int ONES = 512;                          // a constant x record: channel 0 = 1.0 (the bias gradient's A operand), then zeros
  static constexpr int LDS = OFF_ONES + ONES;
  static constexpr int XPIECES = XPATCH / 16 / NT, ZPIECES = ZPATCH / 16 / NT;   // 5, 2
  static constexpr int MTILES = 9 * (C / 32) + 1, MPW = 5;  // 19 M tiles, 5 per wave group (the last group: 4)
  static constexpr int SLAB = (9 * C + 1) * F;              // floats per slab
  static_assert(XPATCH % (16 * NT) == 0 && ZPATCH % (16 * NT) == 0 && LDS <= 160 * 1024 && 4 * MPW >= MTILES, "shape");
};

// four transposing LDS reads from two lane addresses with immediate offsets (issued, not waited for)
#define PCONV_TR4I(o0, o1, o2, o3, a0, a1, a2, a3, imm)                                                                \
  asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%8\n\tds_read_b64_tr_b16 %1, %5 offset:%8\n\t"                          \
               "ds_read_b64_tr_b16 %2, %6 offset:%8\n\tds_read_b64_tr_b16 %3, %7 offset:%8"                              \
               : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)                                                             \
               : "v"((uint32_t)(a0)), "v"((uint32_t)(a1)), "v"((uint32_t)(a2)), "v"((uint32_t)(a3)), "n"(imm)           \
               : "memory")

// the item loop of wave group MG (M tiles 5 MG .. 5 MG + 4, compile-time: every tile's row / parity offset is an immediate)
template <int MG>
__device__ __forceinline__ void pwgradw_items(const PConvWgradArgs& a, unsigned char* p_smem, int z, int it0, int it1, int tid,
                                              int wid, f32x16 (&acc)[PWgradWCfg::MPW][2]) {
  typedef PWgradWCfg G;
  constexpr int R = G::R, NW = G::NW, MPW = G::MPW;
  constexpr int MT0 = MG * MPW, NTILE = MT0 + MPW <= G::MTILES ? MPW : G::MTILES - MT0;   // tiles of this group (5, 5, 5, 4)
  typedef short s16x8_t __attribute__((ext_vector_type(8)));
  const int lane = tid & 63, h = lane >> 5, np = wid & 1;
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3, tr_cb = lane & 16;
  const uint32_t smem_base = (uint32_t)(size_t)(pconv_lptr_t)p_smem;

  // live == false (past the block's last item): the same number of DMA instructions, all from the zero page
  auto dma_x = [&](int item, bool live, unsigned char* buf) {
    const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
    const int oy0 = band * R;
    const char* xsrc = reinterpret_cast<const char*>(a.x[z]) + ((size_t)(b * a.IH + 2 * oy0) * a.IW) * G::XREC;
    const int xrows = live ? min(G::XROWS, a.IH - 2 * oy0) : 0;
    int ln = tid & 63;
    asm volatile("" : "+v"(ln));            // per-lane address terms recomputed per item, not kept in (spilled) registers
#pragma unroll
    for (int i = 0; i < G::XPIECES; ++i) {
      const int l = (i * NW + wid) * 64 + ln;
      const int rec = l >> 4, chp = l & 15;
      const int rr = rec >> 5, par = (rec >> 4) & 1, j = rec & 15;
      const int ix = 2 * j + par;
      const bool ok = rr < xrows && ix < a.IW;
      const char* g = ok ? xsrc + (size_t)(((rr * a.IW + ix) << 4) + (chp ^ pconv_swz<G::XREC>(j))) * 16
                         : reinterpret_cast<const char*>(g_zero_page);
      __builtin_amdgcn_global_load_lds((pconv_gptr_t)g, (pconv_lptr_t)(buf + (i * NW + wid) * 1024), 16, 0, 0);
    }
  };
  auto dma_z = [&](int item, bool live, unsigned char* buf) {
    const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
    const int oy0 = band * R;
    const char* zsrc = reinterpret_cast<const char*>(a.dz[z]) + ((size_t)(b * a.OH + oy0) * a.OW) * G::ZREC;
    const int zrows = live ? a.OH - oy0 : 0;
    int ln = tid & 63;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int i = 0; i < G::ZPIECES; ++i) {
      const int l = (i * NW + wid) * 64 + ln;
      const int rec = l >> 5, chp = l & 31;
      const int row = rec >> 4, ox = rec & 15;
      const bool ok = row < zrows && ox < a.OW;
      const char* g = ok ? zsrc + (size_t)(((row * a.OW + ox) << 5) + (chp ^ (ox & 15))) * 16
                         : reinterpret_cast<const char*>(g_zero_page);
      __builtin_amdgcn_global_load_lds((pconv_gptr_t)g, (pconv_lptr_t)(buf + (i * NW + wid) * 1024), 16, 0, 0);
    }
  };

  // Lane addresses relative to a buffer.  This lane feeds the transposed reads with k-rows (pixels) ox0 and ox0 + 4 and 4
  // consecutive rows (channels tr_cb + 4 tr_p .. + 3 of a 32-channel tile).  x: per (channel half, kx >> 1) the hi-plane
  // addresses of records j0 = ox0 + (kx >> 1) and j0 + 4 in parity region 0 of row 0 (row / parity: immediates); the lo
  // plane is 8 chunks on, i.e. address ^ 128 (the swizzle XORs the chunk bits only).  dz: per N tile the hi-plane addresses of
  // records ox0, ox0 + 4 of row 0; lo plane = address ^ 256.
  const int ox0 = 8 * h + tr_q;
  const int cofs = (tr_cb + 4 * tr_p) * 2;
  uint32_t xa[2][2][2], za[2][2];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int co = 64 * ch + cofs, chunk = co >> 4, within = co & 15;
        const int j = ox0 + kh + 4 * e;
        xa[ch][kh][e] = (uint32_t)(j * G::XREC + within + ((chunk ^ pconv_swz<G::XREC>(j)) << 4));
      }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int co = 64 * (2 * np + t) + cofs, chunk = co >> 4, within = co & 15;
      const int ox = ox0 + 4 * e;
      za[t][e] = (uint32_t)(ox * G::ZREC + within + ((chunk ^ (ox & 15)) << 4));
    }
  const uint32_t ones_addr = smem_base + G::OFF_ONES + cofs;

  // issue order per item: dz of the next item, then x of the one after it; vmcnt retires in order, so "at most the last
  // XPIECES loads in flight" = this item's dz and x have landed, the next item's x may still be on its way
  dma_z(it0, true, p_smem + G::OFF_Z);
  dma_x(it0, true, p_smem);
  dma_x(it0 + 1, it0 + 1 < it1, p_smem + G::XPATCH);
  int xb = 0, zb = 0;
  for (int item = it0; item < it1; ++item) {
    const uint32_t base = smem_base + xb * G::XPATCH, zbase = smem_base + G::OFF_Z + zb * G::ZPATCH;
    static_assert(G::XPIECES == 5, "counted wait");
    asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // this item's bands landed; every wave is done with the previous item's buffers
    const int xn = xb == 0 ? 2 : xb - 1;     // (xb + 2) % 3: the buffer of the previous item
    dma_z(item + 1, item + 1 < it1, p_smem + G::OFF_Z + (zb ^ 1) * G::ZPATCH);
    dma_x(item + 2, item + 2 < it1, p_smem + xn * G::XPATCH);
#pragma unroll
    for (int oyl = 0; oyl < R; ++oyl) {        // k16 step: output row oyl, pixels ox = 0 .. 15
      // The fragments of M tile i + 1 are requested before the MFMAs of tile i (ring of two register sets, counted waits:
      // LDS returns in order).
      s16x4_t bq[2][4], aq[2][4];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint32_t r0 = zbase + za[t][0], r1 = zbase + za[t][1];
        PCONV_TR4I(bq[t][0], bq[t][1], bq[t][2], bq[t][3], r0, r1, r0 ^ 256u, r1 ^ 256u, oyl * G::ZROWB);
      }
#define PWGW_LOAD_A(i, o)                                                                                               \
      {                                                                                                                 \
        constexpr int mt_ = MT0 + (i);                                                                                  \
        if constexpr (mt_ == G::MTILES - 1) {          /* the bias tile: the constant record for every pixel */         \
          PCONV_TR4I(o[0], o[1], o[2], o[3], ones_addr, ones_addr, ones_addr + 256u, ones_addr + 256u, 0);              \
        } else {                                                                                                        \
          constexpr int tap_ = mt_ >> 1, ch_ = mt_ & 1, ky_ = tap_ / 3, kx_ = tap_ - 3 * ky_;                          \
          const uint32_t r0 = base + xa[ch_][kx_ >> 1][0], r1 = base + xa[ch_][kx_ >> 1][1];                            \
          PCONV_TR4I(o[0], o[1], o[2], o[3], r0, r1, r0 ^ 128u, r1 ^ 128u, ((2 * oyl + ky_) * 2 + (kx_ & 1)) * G::XPARB); \
        }                                                                                                               \
      }
      PWGW_LOAD_A(0, aq[0])
      bf16x8_t bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < NTILE; ++i) {
        if (i + 1 < NTILE) {
          if (i == 0) PWGW_LOAD_A(1, aq[1])
          if (i == 1) PWGW_LOAD_A(2, aq[0])
          if (i == 2) PWGW_LOAD_A(3, aq[1])
          if (i == 3) PWGW_LOAD_A(4 < NTILE ? 4 : 0, aq[0])
          asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(aq[i & 1][0]), "+v"(aq[i & 1][1]), "+v"(aq[i & 1][2]), "+v"(aq[i & 1][3]),
                       "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[0][2]), "+v"(bq[0][3]), "+v"(bq[1][0]), "+v"(bq[1][1]), "+v"(bq[1][2]), "+v"(bq[1][3]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(aq[i & 1][0]), "+v"(aq[i & 1][1]), "+v"(aq[i & 1][2]), "+v"(aq[i & 1][3]));
        }
        if (i == 0) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            bh[t] = __builtin_bit_cast(bf16x8_t, (s16x8_t){bq[t][0][0], bq[t][0][1], bq[t][0][2], bq[t][0][3], bq[t][1][0], bq[t][1][1], bq[t][1][2], bq[t][1][3]});
            bl[t] = __builtin_bit_cast(bf16x8_t, (s16x8_t){bq[t][2][0], bq[t][2][1], bq[t][2][2], bq[t][2][3], bq[t][3][0], bq[t][3][1], bq[t][3][2], bq[t][3][3]});
          }
        }
        const s16x4_t* q4 = aq[i & 1];
        // (the bias tile: ah = the ones row, al = 0 -- the exact-zero product in front of ones x lo, ones x hi changes nothing)
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, (s16x8_t){q4[0][0], q4[0][1], q4[0][2], q4[0][3], q4[1][0], q4[1][1], q4[1][2], q4[1][3]});
        const bf16x8_t al = __builtin_bit_cast(bf16x8_t, (s16x8_t){q4[2][0], q4[2][1], q4[2][2], q4[2][3], q4[3][0], q4[3][1], q4[3][2], q4[3][3]});
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[t], acc[i][t], 0, 0, 0);
          acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[t], acc[i][t], 0, 0, 0);
          acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[t], acc[i][t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#undef PWGW_LOAD_A
    }
    xb = xb == 2 ? 0 : xb + 1;
    zb ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the (zero-page) bands requested past the last item
}

__global__ __launch_bounds__(512, 1) void pconv_wgradw_kernel(const PConvWgradArgs a) {
  typedef PWgradWCfg G;
  constexpr int C = G::C, F = G::F, MPW = G::MPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char p_smem[];
  const int z = blockIdx.z;
  const int nitems = a.B * a.nbands;
  const int per = (nitems + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it0 = blockIdx.x * per, it1 = min(nitems, it0 + per);    // the launcher sizes the grid so that it0 < it1
  const int tid = threadIdx.x, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int np = wid & 1, mt0 = (wid >> 1) * MPW;             // N tiles 2 np, 2 np + 1; M tiles mt0 .. mt0 + 4 (< 19)
  // the all-ones row of the bias gradient as data: every pixel of its "tile" reads this record, whose transposed fragment
  // is row 0 = 1.0 for every k (hi plane), zeros elsewhere and in the lo plane
  if (tid < G::ONES / 4) reinterpret_cast<uint32_t*>(p_smem + G::OFF_ONES)[tid] = tid == 0 ? 0x00003F80u : 0u;

  f32x16 acc[MPW][2];
#pragma unroll
  for (int i = 0; i < MPW; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][t][q] = 0.f;
  switch (wid >> 1) {        // one loop per wave group: tile offsets are immediates, register allocation per group
    case 0: pwgradw_items<0>(a, p_smem, z, it0, it1, tid, wid, acc); break;
    case 1: pwgradw_items<1>(a, p_smem, z, it0, it1, tid, wid, acc); break;
    case 2: pwgradw_items<2>(a, p_smem, z, it0, it1, tid, wid, acc); break;
    default: pwgradw_items<3>(a, p_smem, z, it0, it1, tid, wid, acc); break;
  }
  // ---- one slab per block: rows (tap, c) in HWIO order, then the bias row
  float* slab = a.slab + ((size_t)z * gridDim.x + blockIdx.x) * G::SLAB;
#pragma unroll
  for (int i = 0; i < MPW; ++i) {
    const int mt = mt0 + i;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = (q & 3) + 8 * (q >> 2) + 4 * (int)((threadIdx.x >> 5) & 1);
        const int n = (2 * np + t) * 32 + (int)(threadIdx.x & 31);
        if (mt < G::MTILES - 1) slab[(size_t)(mt * 32 + m) * F + n] = acc[i][t][q];
        else if (mt == G::MTILES - 1 && m == 0) slab[(size_t)(9 * C) * F + n] = acc[i][t][q];
      }
  }
}

// as pconv_wgrad_launch: the slab count depends on the geometry and the device only
static int pconv_wgradw_launch(PConvWgradArgs& a, int ngroups, int max_slabs, long long workspace_bytes, int* nslab,
                               hipStream_t st, const char* name) {
  static bool opted[64] = {};
  static int ncu[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), name);
  SERL_CHECK_ARG(dev >= 0 && dev < 64, "%s: device ordinal %d out of range", name, dev);
  if (!opted[dev]) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)pconv_wgradw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PWgradWCfg::LDS), name);
    SERL_CHECK_HIP(hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev), name);
    opted[dev] = true;
  }
  const int nitems = a.B * a.nbands;
  static int cap = -1;                    // SERL_PCONV_WGW_CUS: CUs this kernel spreads over (its slab count: a fixed choice per process)
  if (cap < 0) { const char* e = getenv("SERL_PCONV_WGW_CUS"); cap = e ? atoi(e) : 0; }
  const int cus = ncu[dev] > 0 ? ncu[dev] : 256;
  int gx = (cap > 0 && cap < cus ? cap : cus) / ngroups;
  if (gx > max_slabs) gx = max_slabs;
  if (gx > nitems) gx = nitems;
  if (gx < 1) { *nslab = 0; return SERL_OK; }
  const int per = cdiv(nitems, gx);
  gx = cdiv(nitems, per);                    // every block has at least one item
  if ((long long)ngroups * gx * PWgradWCfg::SLAB * 4 > workspace_bytes) { *nslab = 0; return SERL_OK; }
  *nslab = gx;
  hipLaunchKernelGGL(pconv_wgradw_kernel, dim3(gx, 1, ngroups), dim3(PWgradWCfg::NT), PWgradWCfg::LDS, st, a);
  SERL_CHECK_LAUNCH(name);
  return SERL_OK;
}
