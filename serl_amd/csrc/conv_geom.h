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

// out[g][e] = sum_s slab[g][s][e]; rows < KKC go to dW, the last row to dbias.
struct ReduceArgs {
  const float* slab; float* dw[MAXG]; float* db[MAXG];
  int nsplit, kkc, F;
};
// Fixed-order sum of the split-K slabs (reproducible, no atomics), without LDS and without a barrier: the launches run
// next to persistent patch kernels that hold 150 of a CU's 160 KB of LDS, and a block that needs 4 KB of it found room
// on a CU only one at a time (conv3's 61 MB of slabs: 19 us alone, 84 us inside the update).  A wave = EL = 16 element
// lanes (V consecutive floats each: 256 bytes of one slab row, coalesced) x 4 slab lanes; slab lane sl sums slabs sl,
// sl + 4, ... (independent loads, 8 in flight), then the four partial sums are added across the lanes 16 and 32 apart.
template <int V>
static __global__ __launch_bounds__(256) void wgrad_reduce_kernel(const ReduceArgs a) {
  constexpr int SL = 4, EL = 16;
  const int gi = blockIdx.y;
  const int nel = (a.kkc + 1) * a.F;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ln = lane & (EL - 1), sl = lane / EL;
  const int el = ((blockIdx.x * 4 + wave) * EL + ln) * V;
  const float* base = a.slab + (size_t)gi * a.nsplit * (size_t)nel;
  float s[V];
#pragma unroll
  for (int j = 0; j < V; ++j) s[j] = 0.f;
  if (el < nel) {
#pragma unroll 8
    for (int k = sl; k < a.nsplit; k += SL) {
      const float* src = base + (size_t)k * nel + el;
      if constexpr (V == 4) {
        const float4 v = *reinterpret_cast<const float4*>(src);
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
      } else {
        s[0] += *src;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < V; ++j) {
    s[j] += __shfl_xor(s[j], 16);
    s[j] += __shfl_xor(s[j], 32);
  }
  if (sl == 0 && el < nel) {
    float* dst = el < a.kkc * a.F ? a.dw[gi] + el : a.db[gi] + (el - a.kkc * a.F);
    if constexpr (V == 4) *reinterpret_cast<float4*>(dst) = make_float4(s[0], s[1], s[2], s[3]);
    else *dst = s[0];
  }
}

// Vector variant when F % 4 == 0 (so a float4 never straddles dW / dbias) and every pointer is 16-byte aligned.
static inline void launch_wgrad_reduce(const ReduceArgs& ra, int ngroups, hipStream_t st) {
  const int nel = (ra.kkc + 1) * ra.F;
  bool v4 = ra.F % 4 == 0 && ((uintptr_t)ra.slab & 15) == 0;
  for (int i = 0; i < ngroups; ++i) v4 = v4 && (((uintptr_t)ra.dw[i] | (uintptr_t)ra.db[i]) & 15) == 0;
  if (v4) hipLaunchKernelGGL(wgrad_reduce_kernel<4>, dim3((nel + 255) / 256, ngroups), dim3(256), 0, st, ra);
  else hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((nel + 63) / 64, ngroups), dim3(256), 0, st, ra);
}

#ifndef WGRAD_BLOCKS
#define WGRAD_BLOCKS 1024
#endif
static void wgrad_plan(const ConvGeom& g, int ngroups, int& nsplit, int& klen) {
  const int Kr = g.B * g.OH * g.OW;
  const int kkc1 = g.KH * g.KW * g.C + 1;
  const int bn = g.F <= 32 ? 32 : (g.F <= 64 ? 64 : 128);
  const int tiles = cdiv(kkc1, 128) * cdiv(g.F, bn) * ngroups;
  static const int deep_blocks = [] {
    const char* e = getenv("SERL_AMD_WGRAD_BLOCKS_DEEP");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : WGRAD_BLOCKS;
  }();
  int want = cdiv(kkc1 > 512 ? deep_blocks : WGRAD_BLOCKS, tiles);
  if (want < 1) want = 1;
  klen = cdiv(cdiv(Kr, want), 64) * 64;   // multiple of every kernel's K slab
  if (klen < 64) klen = 64;
  nsplit = cdiv(Kr, klen);
}

