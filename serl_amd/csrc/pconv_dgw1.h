// conv2's data gradient and conv1's weight gradient in ONE kernel (the encoder's first two layers, backward).
//
// conv1 reads the uint8 frames, so nothing needs its data gradient: dz1 = ReLU'(act1) * dgrad(conv2)(dz2) is consumed by
// conv1's weight gradient only.  Written to HBM and read back it is the largest tensor of the backward pass (0.26 GB per
// 2-camera step, a quarter of the phase's traffic, and the uint8 weight-gradient GEMM behind it sat on the main stream's
// critical chain).  Here every 32-pixel x 32-channel tile of dz1 goes from pconv_dgrad_kernel's accumulators straight into
// the weight-gradient product as the MFMA B operand:
//   * the data-gradient half is pconv_dgrad_kernel's (same patches, same k order and MFMA sequence): the tile values are
//     bit-identical to what serl_convs_dgrad stores, including the ReLU byte mask and the hi / lo split;
//   * dW1[(ky, kx, c), f] = 1/255 sum_pixels frame[2 iy + ky, 2 ix + kx, c] * dz1[iy, ix, f]: contraction over the tile's 32
//     pixels = two k16 steps.  A lane of an accumulator tile holds 16 pixels of one channel f, exactly the B-operand shape
//     (the assignment of pixels to k slots is free as long as A agrees).  A operand = the frames: the item's 4P + 1 frame
//     rows are one contiguous span of global memory, brought into LDS by the same double-buffered LDS-DMA as the dz patch;
//     per (ky) block of 32 rows lane r < 27 reads byte (kx, c) = r of the pixel window, 8 pixels per k16 step, and converts
//     them to bf16 (exact).  uint8 x (hi + lo) = 2 MFMAs per block and step, 12 per tile; lane 27 of block ky = 0 reads
//     a constant 1 from LDS: the bias gradient.  Rows 28..31 of a block multiply finite garbage and are never stored.
//   * the three 32 x 32 accumulators (ky) of a wave persist over all items of the block; at the end the eight waves' partial
//     sums are added in wave order through LDS and the block writes one slab (82 x 32, the layout of wgrad_reduce_kernel).
// Classes per wave: waves 0-3 (row pair w) own the parity classes (0,0) and (1,1), waves 4-7 (row pair w - 4) the classes
// (0,1) and (1,0).  Three taps of data-gradient weights per wave live in registers (96); the tap of class (1,1) and the last
// taps of classes (0,0) and (1,0) -- each used once per item and wave -- are read from an LDS image instead, which leaves
// room for the weight-gradient accumulators.  Waves w and w + 4 share a SIMD: 84 + 72 MFMAs per item on each of the four.
#pragma once
#include "pconv.h"

struct PConvDgw1Args {
  const unsigned short* dz[MAXG];     // conv2's dz, S format [B][OH][OW][2][64]
  const unsigned short* wd[MAXG];     // conv2's data-gradient weights, S format [(ky, kx, c)][2][64]
  const unsigned char* xbits[MAXG];   // ReLU byte mask of conv1's output [B*IH*IW][4], or nullptr
  const unsigned char* img[MAXG];     // frames, uint8 [B][IH1][rb1]
  float* slab;                        // [group][gridDim.x][82 * 32]
  int B, IH, IW, OH, OW, nbands;      // IH, IW: conv1's output = conv2's input
  int IH1, rb1;                       // frame rows, bytes per frame row (IW1 * 9)
  float scale;                        // 1 / 255
  FastDiv d_bands;
};

struct PDgw1Cfg {
  static constexpr int C1 = 9, RUN = 3 * C1;                   // bytes (kx, c) of one tap row of a pixel's window
  static constexpr int C = PDgradCfg::C, F = PDgradCfg::F, P = PDgradCfg::P, NT = PDgradCfg::NT;
  static constexpr int PATCH = PDgradCfg::PATCH, NDMA = PDgradCfg::NDMA, MSB = PDgradCfg::MSB, ROWB = PDgradCfg::ROWB;
  static constexpr int BROWS = 4 * P + 1;                      // frame rows of one item
  static constexpr int BANDW = 20;                             // DMA wave-instructions (64 lanes x 16 bytes) per band
  static constexpr int BAND = BANDW * 1024;                    // 20 KB >= BROWS * rb1
  static constexpr int ONES = 1024;                            // constant-1 bytes
  // B fragments read from LDS instead of registers, [tap slot][kc * 2 + plane][lane]: slot 0 = the tap of class (1,1),
  // slot 1 = the last tap of class (0,0), slot 2 = the last tap of class (1,0)
  static constexpr int BTAP = 8 * 64 * 16, BIMG = 3 * BTAP;
  static constexpr int OFF_MS = 2 * PATCH, OFF_BAND = OFF_MS + 2 * MSB, OFF_ONES = OFF_BAND + 2 * BAND, OFF_BIMG = OFF_ONES + ONES;
  static constexpr int LDS = OFF_BIMG + BIMG;
  static constexpr int SLAB = (9 * C1 + 1) * C;                // floats per slab
  static constexpr int RED = 8 * 3 * 16 * 64 * 4;              // the eight waves' accumulators at the end (over the patches)
  static_assert(RUN < 32 && 4 * C1 * 3 + 32 * C1 * 3 + RUN < ONES, "pixel window");
  static_assert(RED <= LDS && LDS <= 160 * 1024, "LDS");
};

typedef __attribute__((address_space(3))) const unsigned char* pconv_lds_u8_t;
typedef __attribute__((address_space(3))) const uint32_t* pconv_lds_u32_t;

// the tile's 12 weight-gradient MFMAs.  acc: lane (f = r, h), q -> pixel p = (q & 3) + 8 (q >> 2) + 4 h of the tile, i.e.
// ix = 2 p + PX.  arow: this lane's byte in the frame band for pixel 0 of a class-(0,0) tile of the wave's row pair.
template <int PY, int PX>
__device__ __forceinline__ void pdgw1_wgrad(pconv_lds_u32_t ms, bool masked, int IW, pconv_lds_u8_t arow, int rb1,
                                            pconv_lds_u8_t ones_ptr, bool ones_lane, int iyl, int r, int h, f32x16& acc,
                                            f32x16* wacc) {
  constexpr int C1 = PDgw1Cfg::C1;
  if (masked) {       // ReLU mask of conv1's output: one dword = the 32 channel bits of a pixel
    pconv_lds_u32_t mrow = ms + ((2 * iyl + PY) * IW + PX + 8 * h);
    const uint32_t bit = 1u << r;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const uint32_t m = mrow[2 * (q & 3) + 16 * (q >> 2)];
      acc[q] = (m & bit) ? acc[q] : 0.f;
    }
  }
  uint32_t hh[8], ll[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split_pair(acc[2 * j], acc[2 * j + 1], hh[j], ll[j]);
  pconv_lds_u8_t ap0 = arow + (2 * PY) * rb1 + 2 * PX * C1;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    __builtin_amdgcn_sched_barrier(0);      // keeps the byte loads of one step (24 registers) from moving above the previous one
    const u32x4_t hv = {hh[4 * s], hh[4 * s + 1], hh[4 * s + 2], hh[4 * s + 3]};
    const u32x4_t lv = {ll[4 * s], ll[4 * s + 1], ll[4 * s + 2], ll[4 * s + 3]};
    const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, hv), bl = __builtin_bit_cast(bf16x8_t, lv);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      if (ky) __builtin_amdgcn_sched_barrier(0);
      pconv_lds_u8_t ap = (ky == 0 && ones_lane) ? ones_ptr : ap0 + ky * rb1;
      u32x4_t pk;
#pragma unroll
      for (int e = 0; e < 4; ++e) {       // k slots 2e, 2e + 1 of this step: pixels (j & 3) + 8 (2 s + (j >> 2)) + 4 h
        const int j0 = 2 * e, j1 = 2 * e + 1;
        const uint32_t f0 = __float_as_uint((float)ap[4 * C1 * (j0 & 3) + 32 * C1 * (2 * s + (j0 >> 2))]);
        const uint32_t f1 = __float_as_uint((float)ap[4 * C1 * (j1 & 3) + 32 * C1 * (2 * s + (j1 >> 2))]);
        pk[e] = __builtin_amdgcn_perm(f1, f0, 0x07060302u);
      }
      const bf16x8_t av = __builtin_bit_cast(bf16x8_t, pk);
      wacc[ky] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bl, wacc[ky], 0, 0, 0);
      wacc[ky] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bh, wacc[ky], 0, 0, 0);
    }
  }
}

// fragments of one tap of the data-gradient weights: lane (r = c, h) holds f = 16 kc + 8 h .. + 7
__device__ __forceinline__ void pdgw1_load_tap(const unsigned short* wd, int tap, int r, int h, bf16x8_t* bh, bf16x8_t* bl) {
  const unsigned short* row = wd + (size_t)((tap * PDgw1Cfg::C + r) * 2) * PDgw1Cfg::F + h * 8;
#pragma unroll
  for (int kc = 0; kc < 4; ++kc) {
    bh[kc] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(row + kc * 16));
    bl[kc] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(row + PDgw1Cfg::F + kc * 16));
  }
}
template <int PY, int PX>
__device__ __forceinline__ constexpr int pdgw1_tap(int t) { return (PY + 2 * (t / (PX ? 1 : 2))) * 3 + (PX + 2 * (t % (PX ? 1 : 2))); }

// pdgrad_tile with the B fragments of the first NREG taps in registers (bh / bl[t * 4 + kc]) and those of the last tap,
// if NREG is one short, read from the LDS image (bimg: this lane's first fragment of the tap's slot).  Same k order.
template <int PY, int PX, int NREG>
__device__ __forceinline__ void pdgw1_tile(const unsigned char* patch, int iyl, int r, int h, const bf16x8_t* bh,
                                           const bf16x8_t* bl, uint32_t bimg, f32x16& acc) {
  constexpr int TY = PY ? 1 : 2, TX = PX ? 1 : 2, NTAP = TY * TX;
  static_assert(NREG == NTAP || NREG == NTAP - 1, "at most one tap from LDS");
  constexpr int ROWB = PDgw1Cfg::ROWB;
  u32x4_t bq[8];
  if constexpr (NREG < NTAP)     // issued here, waited for in front of the tap
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t"
                 "ds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\tds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168"
                 : "=&v"(bq[0]), "=&v"(bq[1]), "=&v"(bq[2]), "=&v"(bq[3]), "=&v"(bq[4]), "=&v"(bq[5]), "=&v"(bq[6]), "=&v"(bq[7])
                 : "v"(bimg) : "memory");
#pragma unroll
  for (int jy = 0; jy < TY; ++jy)
#pragma unroll
    for (int jx = 0; jx < TX; ++jx) {
      const int t = jy * TX + jx;
      if (t) __builtin_amdgcn_sched_barrier(0);   // one tap's A fragments (32 registers) in flight, not the whole tile's
      const int rec = (r - jx) & 31;
      const unsigned char* base = patch + (iyl - jy + 1) * ROWB + rec * 256;
      const int sw = (rec & 15) << 4;
      if (t == NREG)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(bq[4]), "+v"(bq[5]), "+v"(bq[6]), "+v"(bq[7]) :: "memory");
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(base + (((kc * 2 + h) << 4) ^ sw)));
        const bf16x8_t al = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t*>(base + (((kc * 2 + h + 8) << 4) ^ sw)));
        const bf16x8_t wh = t < NREG ? bh[(t < NREG ? t : 0) * 4 + kc] : __builtin_bit_cast(bf16x8_t, bq[2 * kc]);
        const bf16x8_t wl = t < NREG ? bl[(t < NREG ? t : 0) * 4 + kc] : __builtin_bit_cast(bf16x8_t, bq[2 * kc + 1]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, wh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wh, acc, 0, 0, 0);
      }
    }
}

template <bool MASKED>
__global__ __launch_bounds__(512, 1) void pconv_dgw1_kernel(const PConvDgw1Args a) {
  typedef PDgw1Cfg G;
  constexpr int P = G::P, NDMA = G::NDMA;
  extern __shared__ __attribute__((aligned(16))) unsigned char p_smem[];
  const int z = blockIdx.z;
  const int nitems = a.B * a.nbands;
  const int per = (nitems + (int)gridDim.x - 1) / (int)gridDim.x;
  const int it0 = blockIdx.x * per, it1 = min(nitems, it0 + per);    // the launcher sizes the grid so that it0 < it1
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int iyl = wid & 3;                  // row pair of the band
  const bool type_x = wid < 4;              // classes (0,0), (1,1) | classes (0,1), (1,0)
  constexpr bool masked = MASKED;
  const int rb1 = a.rb1;
  unsigned char* const Ms = p_smem + G::OFF_MS;
  unsigned char* const Band = p_smem + G::OFF_BAND;
  const uint32_t bimg = (uint32_t)(size_t)(pconv_lptr_t)(p_smem + G::OFF_BIMG) + lane * 16;

  struct DmaSrc { const char* dz; const char* mk; const char* im; int oy_first, mk_bytes, im_bytes; };
  auto dma_src = [&](int item) {
    const int b = (int)fd_div((uint32_t)item, a.d_bands), band = item - b * a.nbands;
    DmaSrc d;
    d.oy_first = band * P - 1;              // dz row of patch row 0
    d.dz = reinterpret_cast<const char*>(a.dz[z]) + (size_t)b * a.OH * a.OW * PDgradCfg::REC;
    d.mk = reinterpret_cast<const char*>(a.xbits[z]) + ((size_t)(b * a.IH + 2 * P * band) * a.IW) * (G::C / 8);
    d.mk_bytes = max(0, min(2 * P, a.IH - 2 * P * band)) * a.IW * (G::C / 8);
    // frame rows 4P band .. 4P band + 4P of image b: one contiguous span
    d.im = reinterpret_cast<const char*>(a.img[z]) + ((size_t)b * a.IH1 + 4 * P * band) * rb1;
    d.im_bytes = max(0, min(G::BROWS, a.IH1 - 4 * P * band)) * rb1;
    return d;
  };
  auto dma_item = [&](const DmaSrc& d, int buf) {
    unsigned char* patch = p_smem + buf * G::PATCH;
    int lane = tid & 63;
    asm volatile("" : "+v"(lane));          // the per-lane address terms are recomputed per item, not kept in (spilled) registers
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int l = (i * 8 + wid) * 64 + lane;
      const int rec = l >> 4, chp = l & 15;
      const int row = rec >> 5, ox = rec & 31, oy = d.oy_first + row;
      const bool ok = oy >= 0 && oy < a.OH && ox < a.OW;
      const char* g = ok ? d.dz + (size_t)(((oy * a.OW + ox) << 4) + (chp ^ (ox & 15))) * 16
                         : reinterpret_cast<const char*>(g_zero_page);
      __builtin_amdgcn_global_load_lds((pconv_gptr_t)g, (pconv_lptr_t)(patch + (i * 8 + wid) * 1024), 16, 0, 0);
    }
    if (masked) {                           // wave w brings bytes [256 w, 256 w + 256) of the span
      const int off = wid * 256 + lane * 4;
      const char* g = off < d.mk_bytes ? d.mk + off : reinterpret_cast<const char*>(g_zero_page);
      __builtin_amdgcn_global_load_lds((pconv_gptr_t)g, (pconv_lptr_t)(Ms + buf * G::MSB + wid * 256), 4, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < (G::BANDW + 7) / 8; ++i) {
      if (i * 8 + wid < G::BANDW) {         // wave-uniform
        const int off = ((i * 8 + wid) * 64 + lane) * 16;
        const char* g = off < d.im_bytes ? d.im + off : reinterpret_cast<const char*>(g_zero_page);
        __builtin_amdgcn_global_load_lds((pconv_gptr_t)g, (pconv_lptr_t)(Band + buf * G::BAND + (i * 8 + wid) * 1024), 16, 0, 0);
      }
    }
  };
  dma_item(dma_src(it0), 0);

  // constants in LDS: the ones area and the B fragments of the three taps that are not kept in registers
  for (int i = tid; i < G::ONES / 4; i += G::NT) reinterpret_cast<uint32_t*>(p_smem + G::OFF_ONES)[i] = 0x01010101u;
  if (wid < 3) {
    const int tap = wid == 0 ? pdgw1_tap<1, 1>(0) : (wid == 1 ? pdgw1_tap<0, 0>(3) : pdgw1_tap<1, 0>(1));
    bf16x8_t th[4], tl[4];
    pdgw1_load_tap(a.wd[z], tap, r, h, th, tl);
    unsigned char* dst = p_smem + G::OFF_BIMG + wid * G::BTAP + lane * 16;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
      *reinterpret_cast<u32x4_t*>(dst + (2 * kc) * 1024) = __builtin_bit_cast(u32x4_t, th[kc]);
      *reinterpret_cast<u32x4_t*>(dst + (2 * kc + 1) * 1024) = __builtin_bit_cast(u32x4_t, tl[kc]);
    }
  }

  f32x16 wacc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int q = 0; q < 16; ++q) wacc[i][q] = 0.f;
  const bool ones_lane = r == G::RUN;
  const int lane_off = 4 * iyl * rb1 + 16 * G::C1 * h + (r < G::RUN ? r : 0);
  pconv_lds_u8_t const ones_ptr = (pconv_lds_u8_t)(pconv_lptr_t)(p_smem + G::OFF_ONES);

  // one loop per wave type, each with its own register allocation
  auto run = [&](auto tx) {
    constexpr bool X = decltype(tx)::value;
    bf16x8_t bh[12], bl[12];
    if constexpr (X) {
#pragma unroll
      for (int t = 0; t < 3; ++t) pdgw1_load_tap(a.wd[z], pdgw1_tap<0, 0>(t), r, h, bh + 4 * t, bl + 4 * t);
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) pdgw1_load_tap(a.wd[z], pdgw1_tap<0, 1>(t), r, h, bh + 4 * t, bl + 4 * t);
      pdgw1_load_tap(a.wd[z], pdgw1_tap<1, 0>(0), r, h, bh + 8, bl + 8);
    }
    int pb = 0;
    for (int item = it0; item < it1; ++item, pb ^= 1) {
      const unsigned char* patch = p_smem + pb * G::PATCH;
      // this item's patch, mask bytes and frame rows have landed; every wave is done with the other buffers
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (item + 1 < it1) dma_item(dma_src(item + 1), pb ^ 1);
      pconv_lds_u32_t ms = (pconv_lds_u32_t)(pconv_lptr_t)(Ms + pb * G::MSB);
      pconv_lds_u8_t arow = (pconv_lds_u8_t)(pconv_lptr_t)(Band + pb * G::BAND) + lane_off;
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      if constexpr (X) pdgw1_tile<0, 0, 3>(patch, iyl, r, h, bh, bl, bimg + G::BTAP, acc);
      else pdgw1_tile<0, 1, 2>(patch, iyl, r, h, bh, bl, bimg, acc);
      pdgw1_wgrad<0, X ? 0 : 1>(ms, masked, a.IW, arow, rb1, ones_ptr, ones_lane, iyl, r, h, acc, wacc);
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[q] = 0.f;
      if constexpr (X) pdgw1_tile<1, 1, 0>(patch, iyl, r, h, bh, bl, bimg, acc);
      else pdgw1_tile<1, 0, 1>(patch, iyl, r, h, bh + 8, bl + 8, bimg + 2 * G::BTAP, acc);
      pdgw1_wgrad<1, X ? 1 : 0>(ms, masked, a.IW, arow, rb1, ones_ptr, ones_lane, iyl, r, h, acc, wacc);
    }
  };
  if (type_x) run(std::true_type{});
  else run(std::false_type{});

  // ---- one slab per block: the eight waves' accumulators added in wave order
  __syncthreads();
  float* const red = reinterpret_cast<float*>(p_smem);          // [wave][ky][q][lane]
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int q = 0; q < 16; ++q) red[((wid * 3 + ky) * 16 + q) * 64 + lane] = wacc[ky][q];
  __syncthreads();
  float* slab = a.slab + ((size_t)z * gridDim.x + blockIdx.x) * G::SLAB;
  for (int e = tid; e < 3 * 32 * 32; e += G::NT) {
    const int ky = e >> 10, m = (e >> 5) & 31, f = e & 31;
    const int q = (m & 3) + 4 * (m >> 3), hs = (m >> 2) & 1;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += red[((w * 3 + ky) * 16 + q) * 64 + f + 32 * hs];
    if (m < G::RUN) slab[(ky * G::RUN + m) * G::C + f] = s * a.scale;
    else if (ky == 0 && m == G::RUN) slab[9 * G::C1 * G::C + f] = s;
  }
}

static bool pconv_dgw1_fits(int IH1, int IW1, int C1, int F1, int F2) {
  if (C1 != PDgw1Cfg::C1 || F1 != PDgw1Cfg::C || F2 != PDgw1Cfg::F || IH1 < 7 || IW1 < 7) return false;
  const int rb1 = IW1 * C1;
  if (rb1 % 16 != 0 || PDgw1Cfg::BROWS * rb1 > PDgw1Cfg::BAND) return false;
  const int IW = (IW1 - 3) / 2 + 1, OW = (IW - 3) / 2 + 1;
  return IW <= 64 && OW <= 31;
}

// *nslab: slabs per group written (0: the workspace cannot hold them)
static int pconv_dgw1_launch(PConvDgw1Args& a, int ngroups, long long workspace_bytes, int* nslab, hipStream_t st,
                             const char* name) {
  static bool opted[64] = {};
  static int ncu[64] = {};
  int dev = 0;
  SERL_CHECK_HIP(hipGetDevice(&dev), name);
  SERL_CHECK_ARG(dev >= 0 && dev < 64, "%s: device ordinal %d out of range", name, dev);
  if (!opted[dev]) {
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)pconv_dgw1_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PDgw1Cfg::LDS), name);
    SERL_CHECK_HIP(hipFuncSetAttribute((const void*)pconv_dgw1_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PDgw1Cfg::LDS), name);
    SERL_CHECK_HIP(hipDeviceGetAttribute(&ncu[dev], hipDeviceAttributeMultiprocessorCount, dev), name);
    opted[dev] = true;
  }
  const int nitems = a.B * a.nbands;
  // the slab count fixes the summation order: it depends on the geometry and the device only (as pconv_wgrad_launch)
  const int cus = ncu[dev] > 0 ? ncu[dev] : 256;
  int gx = cus / ngroups;
  if (gx < 1) gx = 1;
  if (gx > nitems) gx = nitems;
  gx = cdiv(nitems, cdiv(nitems, gx));       // every block has at least one item
  if ((long long)ngroups * gx * PDgw1Cfg::SLAB * 4 > workspace_bytes) { *nslab = 0; return SERL_OK; }
  *nslab = gx;
  if (a.xbits[0] != nullptr) hipLaunchKernelGGL(pconv_dgw1_kernel<true>, dim3(gx, 1, ngroups), dim3(PDgw1Cfg::NT), PDgw1Cfg::LDS, st, a);
  else hipLaunchKernelGGL(pconv_dgw1_kernel<false>, dim3(gx, 1, ngroups), dim3(PDgw1Cfg::NT), PDgw1Cfg::LDS, st, a);
  SERL_CHECK_LAUNCH(name);
  return SERL_OK;
}
