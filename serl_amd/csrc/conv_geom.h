// Shared by conv.hip (fp32 / in-kernel split) and convs.hip (S-format operands): geometry, the split-K
// plan of the weight gradient and its fixed-order slab reduction.
#pragma once
#include "common.h"

#define MAXG 8

struct ConvGeom {
  int B, IH, IW, C, OH, OW, F, KH, KW, S;
  FastDiv d_ohow, d_ow, d_kwc;
};

static int make_geom(ConvGeom& g, int B, int IH, int IW, int C, int KH, int KW, int S, int F) {
  SERL_CHECK_ARG(B > 0 && IH >= KH && IW >= KW && C > 0 && F > 0 && S > 0 && KH > 0 && KW > 0, "conv: bad geometry");
  g.B = B; g.IH = IH; g.IW = IW; g.C = C; g.KH = KH; g.KW = KW; g.S = S; g.F = F;
  g.OH = (IH - KH) / S + 1;
  g.OW = (IW - KW) / S + 1;
  SERL_CHECK_ARG((long long)B * IH * IW * C < (1ll << 31) && (long long)B * g.OH * g.OW * F < (1ll << 31),
                 "conv: tensor exceeds 2^31 elements");
  g.d_ohow = make_fastdiv(g.OH * g.OW);
  g.d_ow = make_fastdiv(g.OW);
  g.d_kwc = make_fastdiv(KW * C);
  return SERL_OK;
}

// out[g][e] = sum_s slab[g][s][e]; rows < KKC go to dW, the last row to dbias. 16 elements x 16 slab lanes.
struct ReduceArgs {
  const float* slab; float* dw[MAXG]; float* db[MAXG];
  int nsplit, kkc, F;
};
static __global__ __launch_bounds__(256) void wgrad_reduce_kernel(const ReduceArgs a) {
  // 16 elements x 16 slab lanes per block; lane sl sums slabs sl, sl+16, ... (independent loads in flight),
  // then the 16 partials are added in a fixed order.
  __shared__ float part[16][17];
  const int gi = blockIdx.y;
  const int nel = (a.kkc + 1) * a.F;
  const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int el = blockIdx.x * 16 + e;
  const float* base = a.slab + (size_t)gi * a.nsplit * (size_t)nel;
  float s = 0.f;
  if (el < nel) {
#pragma unroll 8
    for (int k = sl; k < a.nsplit; k += 16) s += base[(size_t)k * nel + el];
  }
  part[sl][e] = s;
  __syncthreads();
  if (sl == 0 && el < nel) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += part[k][e];
    if (el < a.kkc * a.F) a.dw[gi][el] = v;
    else a.db[gi][el - a.kkc * a.F] = v;
  }
}

#ifndef WGRAD_BLOCKS
#define WGRAD_BLOCKS 1024
#endif
static void wgrad_plan(const ConvGeom& g, int ngroups, int& nsplit, int& klen) {
  const int Kr = g.B * g.OH * g.OW;
  const int kkc1 = g.KH * g.KW * g.C + 1;
  const int bn = g.F <= 32 ? 32 : (g.F <= 64 ? 64 : 128);
  const int tiles = cdiv(kkc1, 128) * cdiv(g.F, bn) * ngroups;
  int want = cdiv(WGRAD_BLOCKS, tiles);
  if (want < 1) want = 1;
  klen = cdiv(cdiv(Kr, want), 64) * 64;   // multiple of every kernel's K slab
  if (klen < 64) klen = 64;
  nsplit = cdiv(Kr, klen);
}

