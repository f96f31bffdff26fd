// NHWC / HWIO VALID convolution (forward, data-gradient, weight-gradient) as implicit GEMMs
// on the exact-fp32 MFMA pipe.  Follows flax nn.Conv as used by
// serl/networks/encoders/two_d4pg_encoder.py:30-38 (bias, then ReLU).
#include "igemm_bf3.h"

#include "conv_geom.h"

// x/255 correctly rounded (reference: pixel_multiplexer.py:27 divides in float32).
__device__ __forceinline__ float u8_to_unit(unsigned char v) {
  const float f = (float)v;
  const float rcp = 1.0f / 255.0f;
  float q = f * rcp;
  const float rem = fmaf(-q, 255.0f, f);
  return fmaf(rem, rcp, q);
}

template <bool U8, int V>
__device__ __forceinline__ void ld_in(const void* p, uint32_t idx, VecF<V>& o) {
  if constexpr (U8) {
    static_assert(V == 1, "uint8 input is loaded bytewise");
    o.v[0] = u8_to_unit(((const unsigned char*)p)[idx]);
  } else {
    ld_vec<V>((const float*)p + idx, o);
  }
}
template <int V> __device__ __forceinline__ void zero_vec(VecF<V>& o) {
#pragma unroll
  for (int j = 0; j < V; ++j) o.v[j] = 0.f;
}

// ------------------------------------------------------------------------------ forward
template <bool U8, int V>
struct ConvFwdP {
  static constexpr bool A_KCONTIG = true, B_NCONTIG = true;
  static constexpr int VA = V, VB = V;
  ConvGeom g;
  const void* x[MAXG];
  const float* w[MAXG];
  const float* bias[MAXG];
  float* y[MAXG];
  int relu;
  struct Ctx { const void* x; const float* w; const float* bias; float* y; };
  struct ARow { uint32_t base; bool ok; };
  struct AK { uint32_t off; bool ok; };
  struct BCol { uint32_t n; bool ok; };
  struct BK_ { uint32_t off; bool ok; };
  struct ORow { uint32_t base; };
  struct OCol { float bias; };
  __device__ bool init(Ctx& c, int z) const { c.x = x[z]; c.w = w[z]; c.bias = bias[z]; c.y = y[z]; return true; }
  __device__ int M(const Ctx&) const { return g.B * g.OH * g.OW; }
  __device__ int N(const Ctx&) const { return g.F; }
  __device__ int K(const Ctx&) const { return g.KH * g.KW * g.C; }
  __device__ int ksplit_len() const { return 1 << 30; }
  __device__ void a_row(const Ctx& c, int m, ARow& r) const {
    r.ok = m < M(c);
    const uint32_t mm = r.ok ? m : 0;
    const uint32_t b = fd_div(mm, g.d_ohow), rem = mm - b * (g.OH * g.OW);
    const uint32_t oy = fd_div(rem, g.d_ow), ox = rem - oy * g.OW;
    r.base = ((b * g.IH + oy * g.S) * g.IW + ox * g.S) * g.C;
  }
  __device__ void a_k(const Ctx&, int k, int kend, AK& a) const {
    a.ok = k < kend;
    const uint32_t kk = a.ok ? k : 0;
    const uint32_t ky = fd_div(kk, g.d_kwc);
    a.off = ky * (g.IW * g.C) + (kk - ky * (g.KW * g.C));
  }
  __device__ void a_load(const Ctx& c, const ARow& r, const AK& a, VecF<V>& raw, float& mul, VecF<V>& add) const {
    const bool ok = r.ok && a.ok;
    mul = ok ? 1.f : 0.f; zero_vec(add);
    ld_in<U8, V>(c.x, ok ? r.base + a.off : 0u, raw);
  }
  __device__ void b_col(const Ctx&, int n, BCol& b) const { b.ok = n < g.F; b.n = b.ok ? n : 0; }
  __device__ void b_k(const Ctx&, int k, int kend, BK_& b) const { b.ok = k < kend; b.off = b.ok ? k * g.F : 0; }
  __device__ void b_load(const Ctx& c, const BK_& k, const BCol& n, VecF<V>& raw, float& mul) const {
    mul = (k.ok && n.ok) ? 1.f : 0.f;
    ld_vec<V>(c.w + k.off + n.n, raw);
  }
  __device__ void o_col(const Ctx& c, int n, OCol& o) const { o.bias = c.bias[n]; }
  __device__ void o_row(const Ctx&, int m, ORow& o) const { o.base = (uint32_t)m * g.F; }
  __device__ float o_aux(const Ctx&, const ORow&, int, int) const { return 0.f; }
  __device__ void store(const Ctx& c, const ORow& o, const OCol& oc, int n, int, float v, float) const {
    v += oc.bias;
    if (relu) v = fmaxf(v, 0.f);
    c.y[o.base + n] = v;
  }
};

// ------------------------------------------------------------------------------ dgrad
// dX[b,iy,ix,c] = sum_{ky,kx,n} dZ[b,(iy-ky)/S,(ix-kx)/S,n] * W[ky,kx,c,n]; input pixels are split in
// S*S parity classes so that every class is a dense GEMM over its own taps (no wasted MACs).
template <int V>
struct ConvDgradP {
  static constexpr bool A_KCONTIG = true, B_NCONTIG = false;
  static constexpr int VA = V, VB = V;
  ConvGeom g;
  int G;
  const float* dz[MAXG];
  const float* w[MAXG];
  const float* xact[MAXG];  // forward input of this layer (post-ReLU output of the previous): mask
  float* dx[MAXG];
  struct Cls { int py, px, IHc, IWc, TY, TX; FastDiv d_hw, d_w, d_tx; };
  Cls cls[4];   // host-precomputed parity classes (stride <= 2)
  FastDiv d_f_;
  struct Ctx {
    const float* dz; const float* w; const float* xact; float* dx;
    int py, px, IHc, IWc, TY, TX;
    FastDiv d_hw, d_w, d_f, d_tx;
  };
  struct ARow { int base, iyp, ixp; bool ok; };
  struct AK { int off, jy, jx; bool ok; };
  struct BCol { uint32_t off; bool ok; };
  struct BK_ { uint32_t off; bool ok; };
  struct ORow { uint32_t base; };
  struct OCol {};
  __device__ bool init(Ctx& c, int z) const {
    const int ci = z / G, gi = z - ci * G;
    c.dz = dz[gi]; c.w = w[gi]; c.xact = xact[gi]; c.dx = dx[gi];
    const Cls& k = cls[ci];
    c.py = k.py; c.px = k.px; c.IHc = k.IHc; c.IWc = k.IWc; c.TY = k.TY; c.TX = k.TX;
    c.d_hw = k.d_hw; c.d_w = k.d_w; c.d_tx = k.d_tx; c.d_f = d_f_;
    return c.IHc > 0 && c.IWc > 0;
  }
  __device__ int M(const Ctx& c) const { return g.B * c.IHc * c.IWc; }
  __device__ int N(const Ctx&) const { return g.C; }
  __device__ int K(const Ctx& c) const { return c.TY * c.TX * g.F; }
  __device__ int ksplit_len() const { return 1 << 30; }
  __device__ void a_row(const Ctx& c, int m, ARow& r) const {
    r.ok = m < M(c);
    const uint32_t mm = r.ok ? m : 0;
    const uint32_t b = fd_div(mm, c.d_hw), rem = mm - b * (c.IHc * c.IWc);
    const uint32_t iyp = fd_div(rem, c.d_w), ixp = rem - iyp * c.IWc;
    r.iyp = iyp; r.ixp = ixp;
    r.base = (int)(((b * g.OH + iyp) * g.OW + ixp) * g.F);
  }
  __device__ void a_k(const Ctx& c, int k, int kend, AK& a) const {
    a.ok = k < kend;
    const uint32_t kk = a.ok ? k : 0;
    const uint32_t t = fd_div(kk, c.d_f), n = kk - t * g.F;
    const uint32_t jy = fd_div(t, c.d_tx), jx = t - jy * c.TX;
    a.jy = jy; a.jx = jx;
    a.off = (int)n - (int)((jy * g.OW + jx) * g.F);
  }
  __device__ void a_load(const Ctx& c, const ARow& r, const AK& a, VecF<V>& raw, float& mul, VecF<V>& add) const {
    const int oy = r.iyp - a.jy, ox = r.ixp - a.jx;
    const bool ok = r.ok && a.ok && oy >= 0 && oy < g.OH && ox >= 0 && ox < g.OW;
    mul = ok ? 1.f : 0.f; zero_vec(add);
    ld_vec<V>(c.dz + (ok ? r.base + a.off : 0), raw);
  }
  __device__ void b_col(const Ctx&, int n, BCol& b) const { b.ok = n < g.C; b.off = b.ok ? n * g.F : 0; }
  __device__ void b_k(const Ctx& c, int k, int kend, BK_& b) const {
    b.ok = k < kend;
    const uint32_t kk = b.ok ? k : 0;
    const uint32_t t = fd_div(kk, c.d_f), n = kk - t * g.F;
    const uint32_t jy = fd_div(t, c.d_tx), jx = t - jy * c.TX;
    const uint32_t ky = c.py + g.S * jy, kx = c.px + g.S * jx;
    b.off = ((ky * g.KW + kx) * g.C) * g.F + n;
  }
  __device__ void b_load(const Ctx& c, const BK_& k, const BCol& n, VecF<V>& raw, float& mul) const {
    const bool ok = k.ok && n.ok;
    mul = ok ? 1.f : 0.f;
    ld_vec<V>(c.w + (ok ? k.off + n.off : 0), raw);
  }
  __device__ void o_row(const Ctx& c, int m, ORow& o) const {
    const uint32_t b = fd_div((uint32_t)m, c.d_hw), rem = m - b * (c.IHc * c.IWc);
    const uint32_t iyp = fd_div(rem, c.d_w), ixp = rem - iyp * c.IWc;
    o.base = ((b * g.IH + iyp * g.S + c.py) * g.IW + ixp * g.S + c.px) * g.C;
  }
  __device__ void o_col(const Ctx&, int, OCol&) const {}
  __device__ float o_aux(const Ctx& c, const ORow& o, int n, int) const {
    return c.xact != nullptr ? c.xact[n >= 0 ? o.base + n : 0] : 1.f;
  }
  __device__ void store(const Ctx& c, const ORow& o, const OCol&, int n, int, float v, float aux) const {
    c.dx[o.base + n] = aux > 0.f ? v : 0.f;
  }
};

// ------------------------------------------------------------------------------ wgrad
// dW[(ky,kx,c)][n] = sum_{b,oy,ox} X[b,oy*S+ky,ox*S+kx,c] * dZ[b,oy,ox,n]; one extra all-ones row of
// the virtual A yields dbias[n] = sum dZ for free.  The long reduction is split over blockIdx.y into
// slabs that a second kernel sums in a fixed order (bitwise reproducible, no atomics).
template <bool U8, int V>
struct ConvWgradP {
  static constexpr bool A_KCONTIG = false, B_NCONTIG = true;
  static constexpr int VA = V, VB = V;
  ConvGeom g;
  const void* x[MAXG];
  const float* dz[MAXG];
  float* slab;  // [G][nsplit][KKC+1][F]
  int nsplit, klen;
  struct Ctx { const void* x; const float* dz; float* slab; };
  struct ARow { uint32_t off; int kind; };  // kind: 0 invalid, 1 data, 2 ones
  struct AK { uint32_t base; bool ok; };
  struct BCol { uint32_t n; bool ok; };
  struct BK_ { uint32_t off; bool ok; };
  struct ORow { uint32_t base; };
  struct OCol {};
  __device__ bool init(Ctx& c, int z) const {
    c.x = x[z]; c.dz = dz[z];
    c.slab = slab + (size_t)z * nsplit * (size_t)(M(c) * g.F);
    return true;
  }
  __device__ int KKC() const { return g.KH * g.KW * g.C; }
  __device__ int M(const Ctx&) const { return KKC() + 1; }
  __device__ int N(const Ctx&) const { return g.F; }
  __device__ int K(const Ctx&) const { return g.B * g.OH * g.OW; }
  __device__ int ksplit_len() const { return klen; }
  __device__ void a_row(const Ctx&, int m, ARow& r) const {
    const int kkc = KKC();
    r.kind = m < kkc ? 1 : (m == kkc ? 2 : 0);
    const uint32_t kk = r.kind == 1 ? m : 0;
    const uint32_t ky = fd_div(kk, g.d_kwc);
    r.off = ky * (g.IW * g.C) + (kk - ky * (g.KW * g.C));
  }
  __device__ void a_k(const Ctx&, int k, int kend, AK& a) const {
    a.ok = k < kend;
    const uint32_t mm = a.ok ? k : 0;
    const uint32_t b = fd_div(mm, g.d_ohow), rem = mm - b * (g.OH * g.OW);
    const uint32_t oy = fd_div(rem, g.d_ow), ox = rem - oy * g.OW;
    a.base = ((b * g.IH + oy * g.S) * g.IW + ox * g.S) * g.C;
  }
  __device__ void a_load(const Ctx& c, const ARow& r, const AK& a, VecF<V>& raw, float& mul, VecF<V>& add) const {
    const bool ok = a.ok && r.kind == 1;
    mul = ok ? 1.f : 0.f;
    zero_vec(add);
    add.v[0] = (a.ok && r.kind == 2) ? 1.f : 0.f;     // the all-ones row (first of its vector) that yields dbias
    ld_in<U8, V>(c.x, ok ? a.base + r.off : 0u, raw);
  }
  __device__ void b_col(const Ctx&, int n, BCol& b) const { b.ok = n < g.F; b.n = b.ok ? n : 0; }
  __device__ void b_k(const Ctx&, int k, int kend, BK_& b) const { b.ok = k < kend; b.off = b.ok ? (uint32_t)k * g.F : 0; }
  __device__ void b_load(const Ctx& c, const BK_& k, const BCol& n, VecF<V>& raw, float& mul) const {
    mul = (k.ok && n.ok) ? 1.f : 0.f;
    ld_vec<V>(c.dz + k.off + n.n, raw);
  }
  __device__ void o_row(const Ctx&, int m, ORow& o) const { o.base = (uint32_t)m * g.F; }
  __device__ void o_col(const Ctx&, int, OCol&) const {}
  __device__ float o_aux(const Ctx&, const ORow&, int, int) const { return 0.f; }
  __device__ void store(const Ctx& c, const ORow& o, const OCol&, int n, int split, float v, float) const {
    c.slab[(size_t)split * (size_t)((KKC() + 1) * g.F) + o.base + n] = v;
  }
};

// ------------------------------------------------------------------------------ host API
#ifndef CONV_BK
#define CONV_BK 16
#endif
typedef TileCfg<128, 128, CONV_BK, 2, 2> T128x128;
typedef TileCfg<128, 64, CONV_BK, 2, 2> T128x64;
typedef TileCfg<128, 32, CONV_BK, 4, 1> T128x32;
#ifndef CONV_BK3
#define CONV_BK3 16
#endif
typedef TileCfg<128, 128, CONV_BK3, 2, 2> B128x128;   // bf16_3x tiles
typedef TileCfg<128, 64, CONV_BK3, 2, 2> B128x64;
typedef TileCfg<128, 32, CONV_BK3, 4, 1> B128x32;

template <class P>
static int launch_by_n(const P& p, int max_m, int n, int nsplit, int nz, hipStream_t st, const char* name) {
  if (n <= 32) return igemm_launch_prec<P, T128x32, B128x32>(p, max_m, n, nsplit, nz, st, name);
  if (n <= 64) return igemm_launch_prec<P, T128x64, B128x64>(p, max_m, n, nsplit, nz, st, name);
  return igemm_launch_prec<P, T128x128, B128x128>(p, max_m, n, nsplit, nz, st, name);
}

static bool aligned16(const void* const* ps, int n) {
  for (int i = 0; i < n; ++i)
    if (((uintptr_t)ps[i]) & 15) return false;
  return true;
}

extern "C" int serl_conv2d_fwd(int ngroups, const void* const* x, const float* const* w, const float* const* bias,
                               float* const* y, int in_u8, int B, int IH, int IW, int C, int KH, int KW, int stride,
                               int F, int relu, void* stream) {
  SERL_CHECK_ARG(ngroups >= 1 && ngroups <= MAXG, "conv2d_fwd: ngroups must be in [1,%d]", MAXG);
  ConvGeom g;
  if (int e = make_geom(g, B, IH, IW, C, KH, KW, stride, F)) return e;
  const int M = B * g.OH * g.OW;
  hipStream_t st = (hipStream_t)stream;
  const bool v4 = !in_u8 && C % 4 == 0 && F % 4 == 0 && aligned16(x, ngroups) &&
                  aligned16((const void* const*)w, ngroups);
#define FILL(p) p.g = g; p.relu = relu; \
  for (int i = 0; i < ngroups; ++i) { p.x[i] = x[i]; p.w[i] = w[i]; p.bias[i] = bias[i]; p.y[i] = y[i]; }
  if (in_u8) { ConvFwdP<true, 1> p; FILL(p) return launch_by_n(p, M, F, 1, ngroups, st, "conv2d_fwd_u8"); }
  if (v4) { ConvFwdP<false, 4> p; FILL(p) return launch_by_n(p, M, F, 1, ngroups, st, "conv2d_fwd_f32x4"); }
  ConvFwdP<false, 1> p; FILL(p) return launch_by_n(p, M, F, 1, ngroups, st, "conv2d_fwd_f32");
#undef FILL
}

template <int V>
static int dgrad_go(const ConvGeom& g, int ngroups, const float* const* dz, const float* const* w,
                    const float* const* xact, float* const* dx, int stride, hipStream_t st) {
  ConvDgradP<V> p; p.g = g; p.G = ngroups;
  p.d_f_ = make_fastdiv(g.F);
  for (int ci = 0; ci < stride * stride; ++ci) {
    typename ConvDgradP<V>::Cls& k = p.cls[ci];
    k.py = ci / stride; k.px = ci - k.py * stride;
    k.IHc = (g.IH - k.py + stride - 1) / stride;
    k.IWc = (g.IW - k.px + stride - 1) / stride;
    k.TY = (g.KH - k.py + stride - 1) / stride;
    k.TX = (g.KW - k.px + stride - 1) / stride;
    if (k.TY <= 0 || k.TX <= 0) { k.TY = 0; k.TX = 0; }   // class gets no gradient: K = 0 -> zeros
    k.d_hw = make_fastdiv(k.IHc * k.IWc > 0 ? k.IHc * k.IWc : 1);
    k.d_w = make_fastdiv(k.IWc > 0 ? k.IWc : 1);
    k.d_tx = make_fastdiv(k.TX > 0 ? k.TX : 1);
  }
  for (int i = 0; i < ngroups; ++i) { p.dz[i] = dz[i]; p.w[i] = w[i]; p.xact[i] = xact ? xact[i] : nullptr; p.dx[i] = dx[i]; }
  const int max_m = g.B * cdiv(g.IH, stride) * cdiv(g.IW, stride);
  return launch_by_n(p, max_m, g.C, 1, ngroups * stride * stride, st, "conv2d_dgrad");
}

extern "C" int serl_conv2d_dgrad(int ngroups, const float* const* dz, const float* const* w, const float* const* xact,
                                 float* const* dx, int B, int IH, int IW, int C, int KH, int KW, int stride, int F,
                                 void* stream) {
  SERL_CHECK_ARG(ngroups >= 1 && ngroups <= MAXG, "conv2d_dgrad: ngroups must be in [1,%d]", MAXG);
  ConvGeom g;
  if (int e = make_geom(g, B, IH, IW, C, KH, KW, stride, F)) return e;
  SERL_CHECK_ARG(stride <= 2, "conv2d_dgrad: stride must be 1 or 2");
  const bool v4 = F % 4 == 0 && aligned16((const void* const*)dz, ngroups) && aligned16((const void* const*)w, ngroups);
  if (v4) return dgrad_go<4>(g, ngroups, dz, w, xact, dx, stride, (hipStream_t)stream);
  return dgrad_go<1>(g, ngroups, dz, w, xact, dx, stride, (hipStream_t)stream);
}

extern "C" long long serl_conv2d_wgrad_workspace_bytes(int ngroups, int B, int IH, int IW, int C, int KH, int KW,
                                                       int stride, int F) {
  ConvGeom g;
  if (make_geom(g, B, IH, IW, C, KH, KW, stride, F)) return -1;
  int nsplit, klen;
  wgrad_plan(g, ngroups, nsplit, klen);
  return (long long)ngroups * nsplit * (KH * KW * C + 1) * F * 4ll;
}

extern "C" int serl_conv2d_wgrad(int ngroups, const void* const* x, const float* const* dz, float* const* dw,
                                 float* const* dbias, float* workspace, long long workspace_bytes, int in_u8, int B,
                                 int IH, int IW, int C, int KH, int KW, int stride, int F, void* stream) {
  SERL_CHECK_ARG(ngroups >= 1 && ngroups <= MAXG, "conv2d_wgrad: ngroups must be in [1,%d]", MAXG);
  ConvGeom g;
  if (int e = make_geom(g, B, IH, IW, C, KH, KW, stride, F)) return e;
  int nsplit, klen;
  wgrad_plan(g, ngroups, nsplit, klen);
  const int kkc = KH * KW * C;
  const long long need = (long long)ngroups * nsplit * (kkc + 1) * F * 4ll;
  SERL_CHECK_ARG(workspace != nullptr && workspace_bytes >= need, "conv2d_wgrad: workspace too small (%lld < %lld)",
                 workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  int e;
  const bool v4 = !in_u8 && C % 4 == 0 && F % 4 == 0 && aligned16(x, ngroups) &&
                  aligned16((const void* const*)dz, ngroups);
#define FILLW(p) p.g = g; p.slab = workspace; p.nsplit = nsplit; p.klen = klen; \
  for (int i = 0; i < ngroups; ++i) { p.x[i] = x[i]; p.dz[i] = dz[i]; }
  if (in_u8) { ConvWgradP<true, 1> p; FILLW(p) e = launch_by_n(p, kkc + 1, F, nsplit, ngroups, st, "conv2d_wgrad_u8"); }
  else if (v4) { ConvWgradP<false, 4> p; FILLW(p) e = launch_by_n(p, kkc + 1, F, nsplit, ngroups, st, "conv2d_wgrad_f32x4"); }
  else { ConvWgradP<false, 1> p; FILLW(p) e = launch_by_n(p, kkc + 1, F, nsplit, ngroups, st, "conv2d_wgrad_f32"); }
#undef FILLW
  if (e) return e;
  ReduceArgs ra; ra.slab = workspace; ra.nsplit = nsplit; ra.kkc = kkc; ra.F = F;
  for (int i = 0; i < ngroups; ++i) { ra.dw[i] = dw[i]; ra.db[i] = dbias[i]; }
  launch_wgrad_reduce(ra, ngroups, st);
  SERL_CHECK_LAUNCH("wgrad_reduce");
  return SERL_OK;
}
