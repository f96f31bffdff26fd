// The conv encoder at precision "bf16_3x" with S-format tensors (sgemm.h): activations, activation
// gradients and re-laid-out weights live in HBM as bf16 hi / lo planes, so forward, data-gradient and
// weight-gradient are pure bf16-MFMA implicit GEMMs.  Same math as conv.hip (flax nn.Conv VALID + bias +
// ReLU, serl/networks/encoders/two_d4pg_encoder.py:30-38) and the same split-K / fixed-order reduction for
// the weight gradient.  The first layer reads the uint8 pixels directly: integers 0..255 are exact in
// bf16, so its A operand has no lo plane and 1/255 (pixel_multiplexer.py:27) is applied to the accumulator.
#include <mutex>

#include "conv_geom.h"
#include "sgemm.h"
#include "pconv.h"
#include "pconv_dgw1.h"
#include "pconv4.h"
#include "sgemm_ring.h"

typedef unsigned short ushort_t;

static __device__ unsigned long long g_ones_u8 = 1ull;   // bytes (1, 0, 0, 0, 0, 0, 0, 0)

__device__ __forceinline__ const char* zero_page() { return reinterpret_cast<const char*>(g_zero_page); }

__device__ __forceinline__ void split1(float v, ushort_t& hi, ushort_t& lo) {
  uint32_t h, l;
  split_pair(v, 0.f, h, l);
  hi = (ushort_t)h; lo = (ushort_t)l;
}

// ------------------------------------------------------------------------------ forward
// A: KC, rows = output pixels, k = (ky, kx, c)          [uint8: k' = ky*32 + kx*C + c, KW*C <= 32]
// B: KC, rows = output channels, from wf = S-format [F][2][Kp] (k order as A)
template <bool U8>
struct ConvFwdS {
  static constexpr bool A_KC = true, B_KC = true, A_LO = !U8, B_LO = true, A_U8 = U8, S_OUT = true, XCD_SWZ = false, MASK_OUT = true;
  ConvGeom g;
  int Kp;
  float scale;                      // 1/255 for uint8 input, else 1
  FastDiv d_c, d_kw;
  const void* x[MAXG];
  const ushort_t* wf[MAXG];
  const float* bias[MAXG];
  ushort_t* y[MAXG];
  unsigned char* rmask[MAXG];       // optional byte mask of y's non-zero elements (nullptr: none), see sgemm.h MASK_OUT
  int relu;
  struct Ctx { const char* x; const ushort_t* wf; const float* bias; ushort_t* y; unsigned char* rmask; };
  struct ARow { uint32_t base; bool ok; };
  struct AK { uint32_t off; int sh; bool ok; };
  struct BRow { uint32_t base; bool ok; };
  struct BK_ { uint32_t off; bool ok; };
  struct ORow { uint32_t base, pix; };
  struct OCol { float bias; };
  __device__ bool init(Ctx& c, int z) const {
    c.x = (const char*)x[z]; c.wf = wf[z]; c.bias = bias[z]; c.y = y[z]; c.rmask = rmask[z];
    return true;
  }
  __device__ unsigned char* s_mask(const Ctx& c) const { return c.rmask; }
  __device__ uint32_t s_pix(const ORow& o) const { return o.pix; }
  __device__ int M(const Ctx&) const { return g.B * g.OH * g.OW; }
  __device__ int N(const Ctx&) const { return g.F; }
  __device__ int K(const Ctx&) const { return Kp; }
  __device__ int ksplit_len() const { return 1 << 30; }
  __device__ void a_row(const Ctx& c, int m, ARow& r) const {
    r.ok = m < M(c);
    const uint32_t mm = r.ok ? m : 0;
    const uint32_t b = fd_div(mm, g.d_ohow), rem = mm - b * (g.OH * g.OW);
    const uint32_t oy = fd_div(rem, g.d_ow), ox = rem - oy * g.OW;
    const uint32_t pix = (b * g.IH + oy * g.S) * g.IW + ox * g.S;
    r.base = U8 ? pix * g.C : pix * 2 * g.C;
  }
  __device__ void a_k(const Ctx&, int k, int kend, AK& a) const {
    a.ok = k < kend;
    a.sh = 0;
    const uint32_t kk = a.ok ? k : 0;
    if constexpr (U8) {
      const int ky = kk >> 5, j0 = kk & 31, run = g.KW * g.C;
      const int over = j0 + 8 - run;                 // bytes of this chunk past the (kx, c) run of row ky
      a.ok = a.ok && j0 < run;
      const int back = over > 0 ? over : 0;          // read the 8 bytes that END at the run's end and shift
      a.off = ky * (g.IW * g.C) + j0 - back;
      a.sh = 8 * back;
    } else {
      const uint32_t tap = fd_div(kk, d_c), c0 = kk - tap * g.C;
      const uint32_t ky = fd_div(tap, d_kw), kx = tap - ky * g.KW;
      a.off = (ky * g.IW + kx) * 2 * g.C + c0;
    }
  }
  __device__ void a_ptr(const Ctx& c, const ARow& r, const AK& a, const char*& hi, const char*& lo, int& sh) const {
    const bool ok = r.ok && a.ok;
    hi = ok ? c.x + (r.base + a.off) : zero_page();
    lo = hi;
    sh = ok ? a.sh : 0;
  }
  __device__ void a_ptr(const Ctx& c, const ARow& r, const AK& a, const char*& hi, const char*& lo) const {
    const bool ok = r.ok && a.ok;
    hi = ok ? c.x + 2ull * (r.base + a.off) : zero_page();
    lo = hi + 2 * g.C;
  }
  __device__ void b_row(const Ctx&, int n, BRow& b) const { b.ok = n < g.F; b.base = b.ok ? (uint32_t)n * 2 * Kp : 0; }
  __device__ void b_k(const Ctx&, int k, int kend, BK_& b) const { b.ok = k < kend; b.off = b.ok ? k : 0; }
  __device__ void b_ptr(const Ctx& c, const BRow& n, const BK_& k, const char*& hi, const char*& lo) const {
    const bool ok = n.ok && k.ok;
    hi = ok ? (const char*)(c.wf + n.base + k.off) : zero_page();
    lo = hi + 2 * Kp;
  }
  struct SCol { float b[8]; };
  __device__ void s_col(const Ctx& c, int n, SCol& s) const {
#pragma unroll
    for (int j = 0; j < 8; ++j) s.b[j] = c.bias[n + j];
  }
  __device__ void s_finish(const Ctx&, const ORow&, const SCol& s, int, float (&v)[8]) const {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = fmaf(v[j], scale, s.b[j]);
      if (relu) v[j] = fmaxf(v[j], 0.f);
    }
  }
  __device__ ushort_t* s_out(const Ctx& c) const { return c.y; }
  __device__ size_t s_base(const ORow& o) const { return o.base; }
  __device__ void o_col(const Ctx& c, int n, OCol& o) const { o.bias = c.bias[n]; }
  __device__ void o_row(const Ctx&, int m, ORow& o) const { o.base = (uint32_t)m * 2 * g.F; o.pix = (uint32_t)m; }
  __device__ float o_aux(const Ctx&, const ORow&, int, int) const { return 0.f; }
  __device__ void store(const Ctx& c, const ORow& o, const OCol& oc, int n, int, float v, float) const {
    v = fmaf(v, scale, oc.bias);
    if (relu) v = fmaxf(v, 0.f);
    ushort_t h, l;
    split1(v, h, l);
    c.y[o.base + n] = h;
    c.y[o.base + g.F + n] = l;
  }
};

// ------------------------------------------------------------------------------ dgrad
// per parity class of input pixels (conv.hip): A: KC, rows = input pixels of the class, k = (tap, f) from dz (S);
// B: KC, rows = input channels, from wd = S-format of the HWIO kernel viewed as [(ky,kx,c)][2][F].
struct ConvDgradS {
  static constexpr bool A_KC = true, B_KC = true, A_LO = true, B_LO = true, A_U8 = false, S_OUT = true, XCD_SWZ = false, MASK_OUT = false;
  ConvGeom g;
  int G;
  const ushort_t* dz[MAXG];
  const ushort_t* wd[MAXG];
  const ushort_t* xact[MAXG];       // post-ReLU input of the layer (S format; its hi plane is the mask), or with
  int xact_bits;                    // xact_bits: the byte mask written by the forward (ConvFwdS::rmask)
  ushort_t* dx[MAXG];
  struct Cls { int py, px, IHc, IWc, TY, TX; FastDiv d_hw, d_w, d_tx; };
  Cls cls[4];
  FastDiv d_f_;
  struct Ctx {
    const ushort_t* dz; const ushort_t* wd; const ushort_t* xact; ushort_t* dx;
    int py, px, IHc, IWc, TY, TX;
    FastDiv d_hw, d_w, d_f, d_tx;
  };
  struct ARow { int pix, iyp, ixp; bool ok; };
  struct AK { int dpix, jy, jx, f0; bool ok; };
  struct BRow { uint32_t c; bool ok; };
  struct BK_ { uint32_t off; bool ok; };
  struct ORow { uint32_t base, pix; };
  struct OCol {};
  __device__ bool init(Ctx& c, int z) const {
    const int ci = z / G, gi = z - ci * G;
    c.dz = dz[gi]; c.wd = wd[gi]; c.xact = xact[gi]; c.dx = dx[gi];
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
    r.pix = (int)((b * g.OH + iyp) * g.OW + ixp);
  }
  __device__ void a_k(const Ctx& c, int k, int kend, AK& a) const {
    a.ok = k < kend;
    const uint32_t kk = a.ok ? k : 0;
    const uint32_t t = fd_div(kk, c.d_f), f0 = kk - t * g.F;
    const uint32_t jy = fd_div(t, c.d_tx), jx = t - jy * c.TX;
    a.jy = jy; a.jx = jx; a.f0 = f0;
    a.dpix = (int)(jy * g.OW + jx);
  }
  __device__ void a_ptr(const Ctx& c, const ARow& r, const AK& a, const char*& hi, const char*& lo) const {
    const int oy = r.iyp - a.jy, ox = r.ixp - a.jx;
    const bool ok = r.ok && a.ok && oy >= 0 && oy < g.OH && ox >= 0 && ox < g.OW;
    const uint32_t off = (uint32_t)(r.pix - a.dpix) * 2 * g.F + a.f0;
    hi = ok ? (const char*)(c.dz + off) : zero_page();
    lo = hi + 2 * g.F;
  }
  __device__ void b_row(const Ctx&, int n, BRow& b) const { b.ok = n < g.C; b.c = b.ok ? n : 0; }
  __device__ void b_k(const Ctx& c, int k, int kend, BK_& b) const {
    b.ok = k < kend;
    const uint32_t kk = b.ok ? k : 0;
    const uint32_t t = fd_div(kk, c.d_f), f0 = kk - t * g.F;
    const uint32_t jy = fd_div(t, c.d_tx), jx = t - jy * c.TX;
    const uint32_t ky = c.py + g.S * jy, kx = c.px + g.S * jx;
    b.off = ((ky * g.KW + kx) * g.C) * 2 * g.F + f0;
  }
  __device__ void b_ptr(const Ctx& c, const BRow& n, const BK_& k, const char*& hi, const char*& lo) const {
    const bool ok = n.ok && k.ok;
    hi = ok ? (const char*)(c.wd + k.off + n.c * 2 * g.F) : zero_page();
    lo = hi + 2 * g.F;
  }
  __device__ void o_row(const Ctx& c, int m, ORow& o) const {
    const uint32_t b = fd_div((uint32_t)m, c.d_hw), rem = m - b * (c.IHc * c.IWc);
    const uint32_t iyp = fd_div(rem, c.d_w), ixp = rem - iyp * c.IWc;
    o.pix = (b * g.IH + iyp * g.S + c.py) * g.IW + ixp * g.S + c.px;
    o.base = o.pix * 2 * g.C;
  }
  struct SCol {};
  __device__ void s_col(const Ctx&, int, SCol&) const {}
  __device__ void s_finish(const Ctx& c, const ORow& o, const SCol&, int n, float (&v)[8]) const {
    if (c.xact == nullptr) return;
    if (xact_bits) {
      const uint32_t bits = reinterpret_cast<const unsigned char*>(c.xact)[(size_t)o.pix * (g.C >> 3) + (n >> 3)];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (!(bits & (1u << j))) v[j] = 0.f;
      return;
    }
    const u32x4_t m = *reinterpret_cast<const u32x4_t*>(c.xact + o.base + n);     // hi plane of the post-ReLU input
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!(m[j] & 0x7fffu)) v[2 * j] = 0.f;
      if (!(m[j] & 0x7fff0000u)) v[2 * j + 1] = 0.f;
    }
  }
  __device__ ushort_t* s_out(const Ctx& c) const { return c.dx; }
  __device__ size_t s_base(const ORow& o) const { return o.base; }
  __device__ void o_col(const Ctx&, int, OCol&) const {}
  __device__ float o_aux(const Ctx& c, const ORow& o, int n, int) const {
    if (c.xact == nullptr) return 1.f;
    return (c.xact[n >= 0 ? o.base + n : 0] & 0x7fffu) ? 1.f : 0.f;     // hi plane of a post-ReLU activation
  }
  __device__ void store(const Ctx& c, const ORow& o, const OCol&, int n, int, float v, float aux) const {
    ushort_t h, l;
    split1(aux > 0.f ? v : 0.f, h, l);
    c.dx[o.base + n] = h;
    c.dx[o.base + g.C + n] = l;
  }
};

// ------------------------------------------------------------------------------ wgrad
// A: RC, rows = (ky, kx, c) [+ an all-ones row -> dbias], k = output pixels, from x (S or uint8);
// B: RC, rows = output channels, k = output pixels, from dz (S).  Output: fp32 slabs as in conv.hip.
template <bool U8>
struct ConvWgradS {
  static constexpr bool A_KC = false, B_KC = false, A_LO = !U8, B_LO = true, A_U8 = U8, S_OUT = false, XCD_SWZ = true;
  ConvGeom g;
  FastDiv d_c, d_kw;
  const void* x[MAXG];
  const ushort_t* dz[MAXG];
  float* slab;
  int nsplit, klen, Mp, kkc;      // Mp: rows of the virtual A incl. the ones row; kkc = KH*KW*C
  float scale;
  struct Ctx { const char* x; const ushort_t* dz; float* slab; };
  struct ARow { uint32_t off; int sh, kind; };      // kind: 0 zero, 1 data, 2 ones
  struct AK { uint32_t base; bool ok; };
  struct BRow { uint32_t n; bool ok; };
  struct BK_ { uint32_t off; bool ok; };
  struct ORow { int base; };
  struct OCol {};
  __device__ bool init(Ctx& c, int z) const {
    c.x = (const char*)x[z]; c.dz = dz[z];
    c.slab = slab + (size_t)z * nsplit * (size_t)((kkc + 1) * g.F);
    return true;
  }
  __device__ int M(const Ctx&) const { return Mp; }
  __device__ int N(const Ctx&) const { return g.F; }
  __device__ int K(const Ctx&) const { return g.B * g.OH * g.OW; }
  __device__ int ksplit_len() const { return klen; }
  __device__ void a_row(const Ctx&, int m, ARow& r) const {
    r.sh = 0; r.off = 0;
    if constexpr (U8) {
      const int ky = m >> 5, j0 = m & 31, run = g.KW * g.C;
      r.kind = (ky < g.KH && j0 < run) ? 1 : (m == g.KH * 32 ? 2 : 0);
      if (r.kind == 1) {
        const int over = j0 + 8 - run, back = over > 0 ? over : 0;
        r.off = ky * (g.IW * g.C) + j0 - back;
        r.sh = 8 * back;
      }
    } else {
      r.kind = m < kkc ? 1 : (m == kkc ? 2 : 0);
      if (r.kind == 1) {
        const uint32_t tap = fd_div((uint32_t)m, d_c), c0 = m - tap * g.C;
        const uint32_t ky = fd_div(tap, d_kw), kx = tap - ky * g.KW;
        r.off = (ky * g.IW + kx) * 2 * g.C + c0;
      }
    }
  }
  __device__ void a_k(const Ctx&, int k, int kend, AK& a) const {
    a.ok = k < kend;
    const uint32_t mm = a.ok ? k : 0;
    const uint32_t b = fd_div(mm, g.d_ohow), rem = mm - b * (g.OH * g.OW);
    const uint32_t oy = fd_div(rem, g.d_ow), ox = rem - oy * g.OW;
    const uint32_t pix = (b * g.IH + oy * g.S) * g.IW + ox * g.S;
    a.base = U8 ? pix * g.C : pix * 2 * g.C;
  }
  __device__ void a_ptr(const Ctx& c, const ARow& r, const AK& a, const char*& hi, const char*& lo, int& sh) const {
    const bool data = a.ok && r.kind == 1, ones = a.ok && r.kind == 2;
    hi = data ? c.x + (a.base + r.off) : (ones ? (const char*)&g_ones_u8 : zero_page());
    lo = hi;
    sh = data ? r.sh : 0;
  }
  __device__ void a_ptr(const Ctx& c, const ARow& r, const AK& a, const char*& hi, const char*& lo) const {
    const bool data = a.ok && r.kind == 1, ones = a.ok && r.kind == 2;
    hi = data ? c.x + 2ull * (a.base + r.off) : (ones ? (const char*)&g_ones_chunk : zero_page());
    lo = data ? hi + 2 * g.C : zero_page();
  }
  __device__ void b_row(const Ctx&, int n, BRow& b) const { b.ok = n < g.F; b.n = b.ok ? n : 0; }
  __device__ void b_k(const Ctx&, int k, int kend, BK_& b) const { b.ok = k < kend; b.off = b.ok ? (uint32_t)k * 2 * g.F : 0; }
  __device__ void b_ptr(const Ctx& c, const BRow& n, const BK_& k, const char*& hi, const char*& lo) const {
    const bool ok = n.ok && k.ok;
    hi = ok ? (const char*)(c.dz + k.off + n.n) : zero_page();
    lo = hi + 2 * g.F;
  }
  __device__ void o_row(const Ctx&, int m, ORow& o) const {
    if constexpr (U8) {
      const int ky = m >> 5, j = m & 31, run = g.KW * g.C;
      o.base = (ky < g.KH && j < run) ? (ky * run + j) * g.F : (m == g.KH * 32 ? kkc * g.F : -1);
    } else {
      o.base = m * g.F;
    }
  }
  __device__ void o_col(const Ctx&, int, OCol&) const {}
  __device__ float o_aux(const Ctx&, const ORow&, int, int) const { return 0.f; }
  __device__ void store(const Ctx& c, const ORow& o, const OCol&, int n, int split, float v, float) const {
    if (o.base < 0) return;
    // the ones row must not be scaled: its row index is kkc
    const float sc = (o.base == kkc * g.F) ? 1.f : scale;
    c.slab[(size_t)split * (size_t)((kkc + 1) * g.F) + o.base + n] = v * sc;
  }
};

// ------------------------------------------------------------------------------ format kernels
// fp32 [rows][cols] <-> S format
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ in, ushort_t* __restrict__ out, long long rows, int cols) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols) return;
  const long long r = i / cols;
  const int c = (int)(i - r * cols);
  ushort_t h, l;
  split1(in[i], h, l);
  out[(r * 2) * cols + c] = h;
  out[(r * 2 + 1) * cols + c] = l;
}
__global__ __launch_bounds__(256) void merge_kernel(const ushort_t* __restrict__ in, float* __restrict__ out, long long rows, int cols) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols) return;
  const long long r = i / cols;
  const int c = (int)(i - r * cols);
  const float h = __uint_as_float((uint32_t)in[(r * 2) * cols + c] << 16), l = __uint_as_float((uint32_t)in[(r * 2 + 1) * cols + c] << 16);
  out[i] = h + l;
}

// HWIO fp32 kernel -> wf (S format [F][2][Kp], forward B operand) and wd (S format [(ky,kx,c)][2][F], dgrad B operand)
struct PrepArgs {
  const float* w[MAXG]; ushort_t* wf[MAXG]; ushort_t* wd[MAXG];
  int KH, KWC, F, Kp, pad32, wq4;    // wq4: the layer is the last encoder layer's shape -- also write pconv4.h's fragment-ordered image behind wf
};
__global__ __launch_bounds__(256) void prep_weights_kernel(const PrepArgs a) {
  // tile = 32 k' x 32 f: read and the wd write run along f, the wf write runs along k' (transposed through LDS)
  __shared__ unsigned short th[32][33], tl[32][33];
  const int gi = blockIdx.z;
  const int kp0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kp = kp0 + ty + 8 * i, f = f0 + tx;
    int k = kp;
    bool valid = kp < a.Kp && f < a.F;
    if (a.pad32) {
      const int ky = kp >> 5, j = kp & 31;
      valid = valid && j < a.KWC;
      k = ky * a.KWC + j;
    }
    ushort_t h = 0, l = 0;
    if (valid) {
      split1(a.w[gi][(size_t)k * a.F + f], h, l);
      if (a.wd[gi] != nullptr) {
        a.wd[gi][(size_t)(k * 2) * a.F + f] = h;
        a.wd[gi][(size_t)(k * 2 + 1) * a.F + f] = l;
      }
      if (a.wq4) {
        ushort_t* wq = a.wf[gi] + (size_t)2 * a.Kp * a.F;
        const long long i0 = a.wq4 == 1 ? p4_wq_index(k, f, 0) : pfw_wq_index(k, f, 0, a.Kp);
        wq[i0] = h;
        wq[i0 + 512] = l;          // the lo plane's fragment follows the hi plane's (64 lanes x 8)
      }
    }
    th[ty + 8 * i][tx] = h; tl[ty + 8 * i][tx] = l;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = f0 + ty + 8 * i, kp = kp0 + tx;
    if (f < a.F && kp < a.Kp) {
      a.wf[gi][(size_t)(f * 2) * a.Kp + kp] = th[tx][ty + 8 * i];
      a.wf[gi][(size_t)(f * 2 + 1) * a.Kp + kp] = tl[tx][ty + 8 * i];
    }
  }
}

// All layers x groups of an encoder pass in ONE launch (blockIdx.z = job): the four per-layer launches sat between
// the dependent conv kernels of the critical path.
#define PREP_MAX_JOBS 32
struct PrepJobs {
  const float* w[PREP_MAX_JOBS]; ushort_t* wf[PREP_MAX_JOBS]; ushort_t* wd[PREP_MAX_JOBS];
  int KH[PREP_MAX_JOBS], KWC[PREP_MAX_JOBS], F[PREP_MAX_JOBS], Kp[PREP_MAX_JOBS], pad32[PREP_MAX_JOBS], wq4[PREP_MAX_JOBS];
};
__global__ __launch_bounds__(256) void prep_weights_multi_kernel(const PrepJobs a) {
  __shared__ unsigned short th[32][33], tl[32][33];
  const int j = blockIdx.z;
  const int Kp = a.Kp[j], F = a.F[j], KWC = a.KWC[j], pad32 = a.pad32[j];
  const int kp0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
  if (kp0 >= Kp || f0 >= F) return;                       // block-uniform: the grid covers the largest job
  const float* w = a.w[j]; ushort_t* wf = a.wf[j]; ushort_t* wd = a.wd[j];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int kp = kp0 + ty + 8 * i, f = f0 + tx;
    int k = kp;
    bool valid = kp < Kp && f < F;
    if (pad32) {
      const int ky = kp >> 5, jj = kp & 31;
      valid = valid && jj < KWC;
      k = ky * KWC + jj;
    }
    ushort_t h = 0, l = 0;
    if (valid) {
      split1(w[(size_t)k * F + f], h, l);
      if (wd != nullptr) {
        wd[(size_t)(k * 2) * F + f] = h;
        wd[(size_t)(k * 2 + 1) * F + f] = l;
      }
      if (a.wq4[j]) {
        ushort_t* wq = wf + (size_t)2 * Kp * F;
        const long long i0 = a.wq4[j] == 1 ? p4_wq_index(k, f, 0) : pfw_wq_index(k, f, 0, Kp);
        wq[i0] = h;
        wq[i0 + 512] = l;          // the lo plane's fragment follows the hi plane's (64 lanes x 8)
      }
    }
    th[ty + 8 * i][tx] = h; tl[ty + 8 * i][tx] = l;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int f = f0 + ty + 8 * i, kp = kp0 + tx;
    if (f < F && kp < Kp) {
      wf[(size_t)(f * 2) * Kp + kp] = th[tx][ty + 8 * i];
      wf[(size_t)(f * 2 + 1) * Kp + kp] = tl[tx][ty + 8 * i];
    }
  }
}

// ------------------------------------------------------------------------------ host API
// Tiles (measured per layer, tools/bench_conv.py): one LDS stage (two barriers per K-step, half the LDS, more blocks
// per CU) wins everywhere except the 128-wide weight gradients; a 64-deep K slab (full 128-byte lines per row chunk)
// wins where K is long, 32 on the first two forward layers (K = 96 / 288).
typedef TileCfg<128, 128, 64, 2, 2> S128x128k64;
typedef TileCfg<128, 128, 32, 2, 2> S128x128k32;
typedef TileCfg<128, 64, 64, 4, 2> S128x64k64;      // 8 waves
typedef TileCfg<128, 64, 32, 4, 2> S128x64k32;
typedef TileCfg<128, 32, 64, 4, 1> S128x32k64;
typedef TileCfg<128, 32, 32, 4, 1> S128x32k32;
// 256-row forward tiles for the first two layers (two 32-row MFMA tiles per wave): half the blocks, the per-block
// prologue (row contexts, weight tile) amortised over twice the output; measured +1.7 % on the whole update
// (tools/ab_tiles.sh); 512 rows, 256-row data-gradient tiles and 256 x 128 tiles for layers 3-4 measured slower.
typedef TileCfg<256, 64, 32, 4, 2> S256x64k32;      // 8 waves
typedef TileCfg<256, 32, 32, 4, 1> S256x32k32;

#ifndef S_FWD32
#define S_FWD32 S256x32k32
#endif
#ifndef S_FWD64
#define S_FWD64 S256x64k32
#endif
#ifndef S_DG32
#define S_DG32 S128x32k64
#endif
#ifndef S_DG64
#define S_DG64 S128x64k64
#endif
#ifndef S_BIG
#define S_BIG S128x128k64
#endif
#ifndef S_WG32
#define S_WG32 S128x32k64
#endif
// Weight-gradient tiles: 32-deep K slabs in one LDS stage.  Stand-alone the 64-deep / two-stage tiles are a few
// percent faster, but the weight gradients run on the side stream next to the data gradients: the smaller LDS
// footprint lets both kernels' blocks share a CU (+1.9 % on the whole update, tools/ab_tiles.sh).
#ifndef S_WG64
#define S_WG64 S128x64k32
#endif
#ifndef S_WG64_LBUF
#define S_WG64_LBUF 1
#endif
typedef TileCfg<128, 128, 32, 4, 2> S128x128k32w8;   // 8 waves
#ifndef S_WGBIG
#define S_WGBIG S128x128k32w8
#define S_WGBIG_LBUF 1
#endif
#ifndef S_DGBIG
#define S_DGBIG S128x128k32   // last layer's data gradient: 32-deep slab (+0.5 % next to the weight gradient; 64-deep stand-alone)
#endif
enum SKind { S_FWD, S_DGRAD, S_WGRAD };
template <class P>
static int s_launch_by_n(SKind kind, const P& p, int max_m, int n, int nsplit, int nz, hipStream_t st, const char* name) {
  const bool big_m = (long long)cdiv(max_m, 256) * nz >= 512;   // 256-row tiles only when they still fill the chip twice
  if (n <= 32) {
    if (kind == S_FWD && big_m) return sgemm_launch<P, S_FWD32, 1>(p, max_m, n, nsplit, nz, st, name);
    if (kind == S_FWD) return sgemm_launch<P, S128x32k32, 1>(p, max_m, n, nsplit, nz, st, name);
    if (kind == S_DGRAD) return sgemm_launch<P, S_DG32, 1>(p, max_m, n, nsplit, nz, st, name);
    return sgemm_launch<P, S_WG32, 1>(p, max_m, n, nsplit, nz, st, name);
  }
  if (n <= 64) {
    if (kind == S_FWD && big_m) return sgemm_launch<P, S_FWD64, 1>(p, max_m, n, nsplit, nz, st, name);
    if (kind == S_FWD) return sgemm_launch<P, S128x64k32, 1>(p, max_m, n, nsplit, nz, st, name);
    if (kind == S_DGRAD) return sgemm_launch<P, S_DG64, 1>(p, max_m, n, nsplit, nz, st, name);
    return sgemm_launch<P, S_WG64, S_WG64_LBUF>(p, max_m, n, nsplit, nz, st, name);
  }
  if (kind == S_WGRAD) return sgemm_launch<P, S_WGBIG, S_WGBIG_LBUF>(p, max_m, n, nsplit, nz, st, name);
  if (kind == S_DGRAD) return sgemm_launch<P, S_DGBIG, 1>(p, max_m, n, nsplit, nz, st, name);
  return sgemm_launch<P, S_BIG, 1>(p, max_m, n, nsplit, nz, st, name);
}

static bool all_aligned16(const void* const* ps, int n) {
  for (int i = 0; i < n; ++i)
    if (ps[i] == nullptr || (((uintptr_t)ps[i]) & 15)) return false;
  return true;
}

static int s_kp(int in_u8, int KH, int KW, int C) { return in_u8 ? KH * 32 : KH * KW * C; }

// two layer shapes keep a second image of their weights behind wf, in the MFMA-fragment order of the kernel that reads it:
// 1 = 3x3, 128 -> 256 (pconv4.h, pconv4_fwd_kernel), 2 = 3x3, 64 -> 128 (pconv.h, pconv_fwdw_kernel)
static int s_wq4(int in_u8, int KH, int KW, int C, int F) {
  if (in_u8 || KH != 3 || KW != 3) return 0;
  return C == 128 && F == 256 ? 1 : (C == 64 && F == 128 ? 2 : 0);
}
extern "C" long long serl_convs_weight_bytes(int in_u8, int KH, int KW, int C, int F) {
  return 4ll * s_kp(in_u8, KH, KW, C) * F * (s_wq4(in_u8, KH, KW, C, F) ? 2 : 1);
}

extern "C" int serl_split_f32(const float* in, void* out, long long rows, int cols, void* stream) {
  SERL_CHECK_ARG(in && out && rows >= 1 && cols >= 1, "split_f32: bad args");
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)cdivll(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, in, (ushort_t*)out, rows, cols);
  SERL_CHECK_LAUNCH("split_f32");
  return SERL_OK;
}

extern "C" int serl_merge_split(const void* in, float* out, long long rows, int cols, void* stream) {
  SERL_CHECK_ARG(in && out && rows >= 1 && cols >= 1, "merge_split: bad args");
  hipLaunchKernelGGL(merge_kernel, dim3((unsigned)cdivll(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, (const ushort_t*)in, out, rows, cols);
  SERL_CHECK_LAUNCH("merge_split");
  return SERL_OK;
}

extern "C" int serl_convs_prep_weights(int ngroups, const float* const* w, void* const* wf, void* const* wd, int in_u8,
                                       int KH, int KW, int C, int F, void* stream) {
  SERL_CHECK_ARG(ngroups >= 1 && ngroups <= MAXG, "convs_prep_weights: ngroups must be in [1,%d]", MAXG);
  SERL_CHECK_ARG(KH > 0 && KW > 0 && C > 0 && F > 0 && F % 8 == 0, "convs_prep_weights: F must be a multiple of 8");
  SERL_CHECK_ARG(in_u8 ? (KW * C >= 8 && KW * C <= 32) : C % 8 == 0, "convs_prep_weights: need C %% 8 == 0 (or 8 <= KW*C <= 32 for uint8 input)");
  PrepArgs a;
  a.KH = KH; a.KWC = KW * C; a.F = F; a.Kp = s_kp(in_u8, KH, KW, C); a.pad32 = in_u8; a.wq4 = s_wq4(in_u8, KH, KW, C, F);
  for (int i = 0; i < ngroups; ++i) {
    SERL_CHECK_ARG(w[i] && wf[i], "convs_prep_weights: null pointer");
    a.w[i] = w[i]; a.wf[i] = (ushort_t*)wf[i]; a.wd[i] = wd ? (ushort_t*)wd[i] : nullptr;
  }
  hipLaunchKernelGGL(prep_weights_kernel, dim3(cdiv(a.Kp, 32), cdiv(F, 32), ngroups), dim3(256), 0, (hipStream_t)stream, a);
  SERL_CHECK_LAUNCH("convs_prep_weights");
  return SERL_OK;
}

extern "C" int serl_convs_prep_weights_multi(int njobs, const float* const* w, void* const* wf, void* const* wd,
                                             const int* in_u8, const int* KH, const int* KW, const int* C, const int* F,
                                             void* stream) {
  SERL_CHECK_ARG(njobs >= 1 && njobs <= PREP_MAX_JOBS, "convs_prep_weights_multi: njobs must be in [1,%d]", PREP_MAX_JOBS);
  SERL_CHECK_ARG(w && wf && in_u8 && KH && KW && C && F, "convs_prep_weights_multi: null argument table");
  PrepJobs a;
  int gx = 0, gy = 0;
  for (int i = 0; i < njobs; ++i) {
    SERL_CHECK_ARG(KH[i] > 0 && KW[i] > 0 && C[i] > 0 && F[i] > 0 && F[i] % 8 == 0, "convs_prep_weights_multi: F must be a multiple of 8");
    SERL_CHECK_ARG(in_u8[i] ? (KW[i] * C[i] >= 8 && KW[i] * C[i] <= 32) : C[i] % 8 == 0,
                   "convs_prep_weights_multi: need C %% 8 == 0 (or 8 <= KW*C <= 32 for uint8 input)");
    SERL_CHECK_ARG(w[i] && wf[i], "convs_prep_weights_multi: null pointer");
    a.w[i] = w[i]; a.wf[i] = (ushort_t*)wf[i]; a.wd[i] = wd ? (ushort_t*)wd[i] : nullptr;
    a.KH[i] = KH[i]; a.KWC[i] = KW[i] * C[i]; a.F[i] = F[i]; a.Kp[i] = s_kp(in_u8[i], KH[i], KW[i], C[i]); a.pad32[i] = in_u8[i];
    a.wq4[i] = s_wq4(in_u8[i], KH[i], KW[i], C[i], F[i]);
    gx = a.Kp[i] > gx ? a.Kp[i] : gx; gy = F[i] > gy ? F[i] : gy;
  }
  hipLaunchKernelGGL(prep_weights_multi_kernel, dim3(cdiv(gx, 32), cdiv(gy, 32), njobs), dim3(256), 0, (hipStream_t)stream, a);
  SERL_CHECK_LAUNCH("convs_prep_weights_multi");
  return SERL_OK;
}

// The patch kernels (pconv.h) are chosen wherever the layer fits them; 0 keeps every layer on the im2col kernels (A/B
// measurements and the bit-identity test of the two paths).  Process-wide.
// enable: 0 = im2col kernels only, 1 (default) = patch kernels incl. the wide one (pconv_fwdw_kernel), 2 = the 8-wave
// patch kernels only, 3 = as 1 plus the LDS-DMA ring GEMM (sgemm_ring.h, also bit-identical) for the deep layers no patch
// kernel takes.  The ring GEMM is 5-8 % faster than the register-staged kernel stand-alone but takes the CU's whole LDS
// (one block per CU), so nothing of the other streams runs next to it: -1 % on the whole update, hence not the default.
// 4 = as 1 with the one-set fused layers-1+2 kernel (pconv_fused12_kernel) instead of the two-set one.
// 5 = as 1 without the fused conv2-data-gradient + conv1-weight-gradient kernel (serl_convs_dgrad_wgrad1_supported says 0).
// 7 = as 1 with conv3's weight gradient on the im2col kernel.
// 6 = as 1 without the layer-1-keeping variant of the fused forward (serl_convs_fwd12_keep_supported says 0).
// 8 = as 1 with the last layer's forward (15 x 15 x 128 -> 7 x 7 x 256) on the im2col kernel instead of pconv4_fwd_kernel.
static int g_patch_kernels = 1, g_ring_kernels = 0, g_wide_kernels = 1, g_f12_two_sets = 1, g_dgw1 = 1, g_f12_keep = 1, g_wgradw = 1, g_fwd4 = 1;
extern "C" int serl_convs_set_patch_kernels(int enable) {
  g_patch_kernels = enable ? 1 : 0;
  g_ring_kernels = enable == 3 ? 1 : 0;
  g_wide_kernels = enable == 1 || enable >= 3 ? 1 : 0;
  g_fwd4 = enable == 8 ? 0 : 1;
  g_f12_two_sets = enable == 4 ? 0 : 1;
  g_dgw1 = enable == 5 ? 0 : 1;
  g_f12_keep = enable == 6 ? 0 : 1;
  g_wgradw = enable == 7 ? 0 : 1;
  return SERL_OK;
}

// Persistent patch kernels launched after this call spread over at most `cus` CUs (0: all).  The driver brackets the
// actor's run-ahead pass with it: its persistent blocks otherwise hold every CU for ~100 us at a time, and the short
// kernels of the critic step's latency-bound chain wait for one to retire.  Per calling thread (an acting thread that
// launches patch kernels concurrently keeps its own, unset, limit); the weight gradient's grid ignores it.
extern "C" int serl_convs_set_cu_limit(int cus) {
  g_pconv_cu_limit = cus > 0 ? cus : 0;
  return SERL_OK;
}

#ifndef P4_PF
#define P4_PF 8
#endif
#ifndef S_RING_FWD
#define S_RING_FWD RingCfg<256, 128, 4, 2, 3>     // 8 waves, 64 x 64 per wave, three K slabs of 32 in flight
#endif
#ifndef S_RING_FWD256
#define S_RING_FWD256 RingCfg<128, 256, 2, 4, 3>  // N = 256 in one tile: the im2col operand is fetched once, not per column tile
#endif

extern "C" int serl_convs_fwd(int ngroups, const void* const* x, const void* const* wf, const float* const* bias,
                              void* const* y, int in_u8, int B, int IH, int IW, int C, int KH, int KW, int stride, int F,
                              int relu, void* const* rmask, void* stream) {
  SERL_CHECK_ARG(ngroups >= 1 && ngroups <= MAXG, "convs_fwd: ngroups must be in [1,%d]", MAXG);
  ConvGeom g;
  if (int e = make_geom(g, B, IH, IW, C, KH, KW, stride, F)) return e;
  SERL_CHECK_ARG(F % 8 == 0 && (in_u8 ? (KW * C >= 8 && KW * C <= 32) : C % 8 == 0), "convs_fwd: need F %% 8 == 0 and C %% 8 == 0 (uint8 input: 8 <= KW*C <= 32)");
  SERL_CHECK_ARG(all_aligned16(wf, ngroups) && all_aligned16(y, ngroups) && (in_u8 || all_aligned16(x, ngroups)),
                 "convs_fwd: S-format tensors must be 16-byte aligned");
  const int M = B * g.OH * g.OW;
  hipStream_t st = (hipStream_t)stream;
  if (g_patch_kernels && !in_u8 && KH == 3 && KW == 3 && stride == 2) {
    // 3x3 stride-2 layers whose row fits the patch layout: input rows staged once in LDS (pconv.h), bit-identical output
    const bool fit32 = C == 32 && F == 64 && g.OW <= 32 && (IW + 1) / 2 <= 32;
    // C = 64 -> 128 at OW <= 16: four waves with the whole weight matrix in their registers (pconv_fwdw_kernel)
    const bool fit16 = g_wide_kernels && C == 64 && F == 128 && g.OW <= 16 && (IW + 1) / 2 <= 16;
    if (g_wide_kernels && g_fwd4 && pconv4_fits(B, IH, IW, C, KH, KW, stride, F)) {
      PConv4Args a;
      a.B = B; a.relu = relu;
      for (int i = 0; i < ngroups; ++i) {
        a.x[i] = (const ushort_t*)x[i]; a.wq[i] = (const ushort_t*)wf[i] + (size_t)2 * 9 * C * F; a.bias[i] = bias[i];
        a.y[i] = (ushort_t*)y[i]; a.rmask[i] = rmask ? (unsigned char*)rmask[i] : nullptr;
      }
      return pconv4_fwd_launch<P4Cfg<P4_PF>>(a, ngroups, st, "convs_fwd(last layer)");
    }
    if (fit16) {
      PConvFwdArgs a;
      a.B = B; a.IH = IH; a.IW = IW; a.OH = g.OH; a.OW = g.OW; a.relu = relu; a.wfrag = 1;
      a.nbands = cdiv(g.OH, 4);
      a.d_iw = make_fastdiv(IW); a.d_bands = make_fastdiv(a.nbands);
      for (int i = 0; i < ngroups; ++i) {
        a.x[i] = (const ushort_t*)x[i]; a.wf[i] = (const ushort_t*)wf[i]; a.bias[i] = bias[i]; a.y[i] = (ushort_t*)y[i];
        a.rmask[i] = rmask ? (unsigned char*)rmask[i] : nullptr;
      }
      return pconv_fwdw_launch<PFwdWCfg<64, 128, 4, 16>>(a, ngroups, st, "convs_fwd(patch, wide)");
    }
    if (fit32) {
      PConvFwdArgs a;
      a.B = B; a.IH = IH; a.IW = IW; a.OH = g.OH; a.OW = g.OW; a.relu = relu;
      a.nbands = cdiv(g.OH, fit32 ? 4 : 3);
      a.d_iw = make_fastdiv(IW); a.d_bands = make_fastdiv(a.nbands);
      for (int i = 0; i < ngroups; ++i) {
        a.x[i] = (const ushort_t*)x[i]; a.wf[i] = (const ushort_t*)wf[i]; a.bias[i] = bias[i]; a.y[i] = (ushort_t*)y[i];
        a.rmask[i] = rmask ? (unsigned char*)rmask[i] : nullptr;
      }
      return pconv_fwd_launch<PFwdCfg<32, 64, 4, 32>>(a, ngroups, st, "convs_fwd(patch)");
    }
  }
#define FILL(p)                                                                                              \
  p.g = g; p.relu = relu; p.Kp = s_kp(in_u8, KH, KW, C); p.scale = in_u8 ? 1.0f / 255.0f : 1.0f;             \
  p.d_c = make_fastdiv(C); p.d_kw = make_fastdiv(KW);                                                        \
  for (int i = 0; i < ngroups; ++i) {                                                                        \
    p.x[i] = x[i]; p.wf[i] = (const ushort_t*)wf[i]; p.bias[i] = bias[i]; p.y[i] = (ushort_t*)y[i];          \
    p.rmask[i] = rmask ? (unsigned char*)rmask[i] : nullptr;                                                 \
  }
  if (in_u8) { ConvFwdS<true> p; FILL(p) return s_launch_by_n(S_FWD, p, M, F, 1, ngroups, st, "convs_fwd_u8"); }
  ConvFwdS<false> p; FILL(p)
  if (g_ring_kernels && F > 128 && p.Kp >= 256) return sgemm_ring_launch<ConvFwdS<false>, S_RING_FWD256>(p, M, F, ngroups, st, "convs_fwd(ring)");
  if (g_ring_kernels && F == 128 && p.Kp >= 256) return sgemm_ring_launch<ConvFwdS<false>, S_RING_FWD>(p, M, F, ngroups, st, "convs_fwd(ring)");
  return s_launch_by_n(S_FWD, p, M, F, 1, ngroups, st, "convs_fwd");
#undef FILL
}

// Layers 1 + 2 of a no-gradient encoder pass in one launch (pconv.h, pconv_fused12_kernel): uint8 frames -> 3x3/2 conv
// (-> 32) + bias + ReLU -> 3x3/2 conv (-> 64) + bias [+ ReLU]; layer 1's output never reaches HBM.  Bit-identical to
// serl_convs_fwd(in_u8 = 1) followed by serl_convs_fwd.  serl_convs_fwd12_supported says whether a geometry fits.
static bool fused12_fits(int IH1, int IW1, int C1, int F1, int F2) {
  if (IH1 < 7 || IW1 < 7 || 3 * C1 < 8 || 3 * C1 > 32 || F1 != 32 || F2 != 64) return false;   // (uint8 runs of 8..32 bytes, as serl_convs_fwd)
  const int rb1 = IW1 * C1;
  if (rb1 % 16 != 0 || rb1 > PFusedCfg::RB1MAX) return false;
  const int IW = (IW1 - 3) / 2 + 1, OW = (IW - 3) / 2 + 1;
  return IW <= 64 && OW <= 32 && (IW + 1) / 2 <= 32;
}
extern "C" int serl_convs_fwd12_supported(int IH1, int IW1, int C1, int F1, int F2) {
  return fused12_fits(IH1, IW1, C1, F1, F2) ? 1 : 0;
}
static int convs_fwd12_impl(const char* name, int ngroups, const void* const* x_u8, const void* const* wf1,
                            const float* const* bias1, const void* const* wf2, const float* const* bias2, void* const* y,
                            int B, int IH1, int IW1, int C1, int F1, int F2, int relu, void* const* rmask, void* const* y1,
                            void* const* rmask1, void* stream) {
  SERL_CHECK_ARG(ngroups >= 1 && ngroups <= MAXG, "%s: ngroups must be in [1,%d]", name, MAXG);
  SERL_CHECK_ARG(B > 0 && fused12_fits(IH1, IW1, C1, F1, F2),
                 "%s: geometry %dx%dx%d -> %d -> %d does not fit the fused kernel (serl_convs_fwd12_supported)", name,
                 IH1, IW1, C1, F1, F2);
  SERL_CHECK_ARG((long long)B * IH1 * IW1 * C1 < (1ll << 31), "%s: tensor exceeds 2^31 elements", name);
  SERL_CHECK_ARG(all_aligned16(x_u8, ngroups) && all_aligned16(wf1, ngroups) && all_aligned16(wf2, ngroups) &&
                 all_aligned16(y, ngroups), "%s: frames, weights and output must be 16-byte aligned", name);
  if (y1 != nullptr) {
    bool ok = g_f12_two_sets != 0;
    for (int i = 0; i < ngroups; ++i) ok = ok && (((uintptr_t)y1[i]) & 15) == 0;     // NULL: that group keeps nothing
    SERL_CHECK_ARG(ok, "%s: keeping layer 1's output needs the two-set kernel (serl_convs_fwd12_keep_supported) and 16-byte aligned buffers", name);
  }
  PConvFusedArgs a;
  a.B = B; a.IH1 = IH1; a.IW1 = IW1; a.C1 = C1; a.relu = relu;
  a.IH = (IH1 - 3) / 2 + 1;
  a.IW = (IW1 - 3) / 2 + 1;
  a.OH = (a.IH - 3) / 2 + 1; a.OW = (a.IW - 3) / 2 + 1;
  a.nbands = cdiv(a.OH, g_f12_two_sets ? PFusedPPCfg::R : PFusedCfg::R);
  a.d_bands = make_fastdiv(a.nbands);
  for (int i = 0; i < ngroups; ++i) {
    SERL_CHECK_ARG(bias1[i] && bias2[i], "%s: null bias", name);
    a.u8[i] = (const unsigned char*)x_u8[i]; a.wf1[i] = (const ushort_t*)wf1[i]; a.bias1[i] = bias1[i];
    a.wf2[i] = (const ushort_t*)wf2[i]; a.bias2[i] = bias2[i]; a.y[i] = (ushort_t*)y[i];
    a.rmask[i] = rmask ? (unsigned char*)rmask[i] : nullptr;
    a.y1[i] = y1 ? (ushort_t*)y1[i] : nullptr;
    a.rmask1[i] = rmask1 ? (unsigned char*)rmask1[i] : nullptr;
    SERL_CHECK_ARG(a.y1[i] != nullptr || a.rmask1[i] == nullptr, "%s: a layer-1 mask needs the layer-1 output buffer of that group", name);
  }
  if (g_f12_two_sets) return pconv_fused12pp_launch(a, ngroups, (hipStream_t)stream, "convs_fwd12(two sets)");
  return pconv_fused12_launch(a, ngroups, (hipStream_t)stream, "convs_fwd12");
}
extern "C" int serl_convs_fwd12(int ngroups, const void* const* x_u8, const void* const* wf1, const float* const* bias1,
                                const void* const* wf2, const float* const* bias2, void* const* y, int B, int IH1,
                                int IW1, int C1, int F1, int F2, int relu, void* const* rmask, void* stream) {
  return convs_fwd12_impl("convs_fwd12", ngroups, x_u8, wf1, bias1, wf2, bias2, y, B, IH1, IW1, C1, F1, F2, relu, rmask,
                          nullptr, nullptr, stream);
}
// serl_convs_fwd12 for a pass whose backward follows: the groups with y1[i] != NULL also get layer 1's output (S format
// [B][IH][IW][32], what serl_convs_fwd(in_u8 = 1) stores -- rows 0 .. 2 OH, the ones layer 2 reads) and, with rmask1[i], its
// ReLU byte mask, written from the kernel's LDS patch: layer 1's output is written once and not read back by layer 2.
extern "C" int serl_convs_fwd12_keep_supported(int IH1, int IW1, int C1, int F1, int F2) {
  return g_patch_kernels && g_f12_two_sets && g_f12_keep && fused12_fits(IH1, IW1, C1, F1, F2) ? 1 : 0;
}
extern "C" int serl_convs_fwd12_keep(int ngroups, const void* const* x_u8, const void* const* wf1,
                                     const float* const* bias1, const void* const* wf2, const float* const* bias2,
                                     void* const* y, int B, int IH1, int IW1, int C1, int F1, int F2, int relu,
                                     void* const* rmask, void* const* y1, void* const* rmask1, void* stream) {
  return convs_fwd12_impl("convs_fwd12_keep", ngroups, x_u8, wf1, bias1, wf2, bias2, y, B, IH1, IW1, C1, F1, F2, relu,
                          rmask, y1, rmask1, stream);
}

extern "C" int serl_convs_dgrad(int ngroups, const void* const* dz, const void* const* wd, const void* const* xact,
                                void* const* dx, int B, int IH, int IW, int C, int KH, int KW, int stride, int F,
                                int xact_bits, void* stream) {
  SERL_CHECK_ARG(ngroups >= 1 && ngroups <= MAXG, "convs_dgrad: ngroups must be in [1,%d]", MAXG);
  ConvGeom g;
  if (int e = make_geom(g, B, IH, IW, C, KH, KW, stride, F)) return e;
  SERL_CHECK_ARG(stride <= 2, "convs_dgrad: stride must be 1 or 2");
  SERL_CHECK_ARG(F % 8 == 0 && C % 8 == 0, "convs_dgrad: need C %% 8 == 0 and F %% 8 == 0");
  SERL_CHECK_ARG(all_aligned16(dz, ngroups) && all_aligned16(wd, ngroups) && all_aligned16(dx, ngroups),
                 "convs_dgrad: S-format tensors must be 16-byte aligned");
  const bool dg32 = C == 32 && F == 64 && g.OW <= 31 && IW <= 64;
  // C = 64 <- F = 128 at OW <= 15: four waves with the weights of all four parity classes in their registers
  const bool dg64 = g_wide_kernels && C == 64 && F == 128 && g.OW <= 15 && IW <= 32;
  if (g_patch_kernels && KH == 3 && KW == 3 && stride == 2 && (dg32 || dg64) && (xact == nullptr || xact_bits)) {
    // dz rows staged once in LDS, all four parity classes computed from there (pconv.h); bit-identical output
    bool uniform = true;                // the byte mask is given for every group or for none
    if (xact != nullptr)
      for (int i = 0; i < ngroups; ++i) uniform = uniform && ((xact[i] != nullptr) == (xact[0] != nullptr));
    if (uniform) {
      PConvDgradArgs a;
      a.B = B; a.IH = IH; a.IW = IW; a.OH = g.OH; a.OW = g.OW;
      static_assert(PDgradCfg::P == PDgradWCfg::P, "band height");
      a.nbands = cdiv((IH + 1) / 2, PDgradCfg::P);
      a.d_bands = make_fastdiv(a.nbands);
      for (int i = 0; i < ngroups; ++i) {
        a.dz[i] = (const ushort_t*)dz[i]; a.wd[i] = (const ushort_t*)wd[i]; a.dx[i] = (ushort_t*)dx[i];
        a.xbits[i] = xact ? (const unsigned char*)xact[i] : nullptr;
      }
      if (dg64) return pconv_dgradw_launch(a, ngroups, (hipStream_t)stream, "convs_dgrad(patch, wide)");
      return pconv_dgrad_launch(a, ngroups, (hipStream_t)stream, "convs_dgrad(patch)");
    }
  }
  ConvDgradS p; p.g = g; p.G = ngroups; p.xact_bits = xact_bits;
  p.d_f_ = make_fastdiv(g.F);
  for (int ci = 0; ci < stride * stride; ++ci) {
    ConvDgradS::Cls& k = p.cls[ci];
    k.py = ci / stride; k.px = ci - k.py * stride;
    k.IHc = (g.IH - k.py + stride - 1) / stride;
    k.IWc = (g.IW - k.px + stride - 1) / stride;
    k.TY = (g.KH - k.py + stride - 1) / stride;
    k.TX = (g.KW - k.px + stride - 1) / stride;
    if (k.TY <= 0 || k.TX <= 0) { k.TY = 0; k.TX = 0; }
    k.d_hw = make_fastdiv(k.IHc * k.IWc > 0 ? k.IHc * k.IWc : 1);
    k.d_w = make_fastdiv(k.IWc > 0 ? k.IWc : 1);
    k.d_tx = make_fastdiv(k.TX > 0 ? k.TX : 1);
  }
  for (int i = 0; i < ngroups; ++i) {
    p.dz[i] = (const ushort_t*)dz[i]; p.wd[i] = (const ushort_t*)wd[i];
    p.xact[i] = xact ? (const ushort_t*)xact[i] : nullptr; p.dx[i] = (ushort_t*)dx[i];
  }
  const int max_m = g.B * cdiv(g.IH, stride) * cdiv(g.IW, stride);
  return s_launch_by_n(S_DGRAD, p, max_m, g.C, 1, ngroups * stride * stride, (hipStream_t)stream, "convs_dgrad");
}

// conv2's data gradient and conv1's weight + bias gradient in one launch (pconv_dgw1.h): dz1 = mask * dgrad(dz2) is used as
// it leaves the accumulators and never reaches HBM.  Replaces serl_convs_dgrad(layer 2) + serl_convs_wgrad(in_u8 = 1,
// layer 1): same dz1 bits, dW1 / dbias1 equal to the separate kernels' up to the order of the fp32 sums.
extern "C" int serl_convs_dgrad_wgrad1_supported(int IH1, int IW1, int C1, int F1, int F2) {
  return g_patch_kernels && g_dgw1 && pconv_dgw1_fits(IH1, IW1, C1, F1, F2) ? 1 : 0;
}
extern "C" long long serl_convs_dgrad_wgrad1_workspace_bytes(int ngroups) {
  return (long long)ngroups * 1024 * PDgw1Cfg::SLAB * 4;      // one slab per block, at most 1024 blocks per group
}
extern "C" int serl_convs_dgrad_wgrad1(int ngroups, const void* const* dz2, const void* const* wd2,
                                       const void* const* rmask1, const void* const* frames_u8, float* const* dw1,
                                       float* const* dbias1, float* workspace, long long workspace_bytes, int B, int IH1,
                                       int IW1, int C1, int F1, int F2, void* stream) {
  SERL_CHECK_ARG(ngroups >= 1 && ngroups <= MAXG, "convs_dgrad_wgrad1: ngroups must be in [1,%d]", MAXG);
  SERL_CHECK_ARG(B > 0 && pconv_dgw1_fits(IH1, IW1, C1, F1, F2),
                 "convs_dgrad_wgrad1: geometry %dx%dx%d -> %d -> %d does not fit the fused kernel (serl_convs_dgrad_wgrad1_supported)",
                 IH1, IW1, C1, F1, F2);
  SERL_CHECK_ARG((long long)B * IH1 * IW1 * C1 < (1ll << 31), "convs_dgrad_wgrad1: tensor exceeds 2^31 elements");
  SERL_CHECK_ARG(all_aligned16(dz2, ngroups) && all_aligned16(wd2, ngroups) && all_aligned16(frames_u8, ngroups),
                 "convs_dgrad_wgrad1: dz, weights and frames must be 16-byte aligned");
  SERL_CHECK_ARG(workspace != nullptr, "convs_dgrad_wgrad1: workspace missing");
  bool uniform = true;                  // the byte mask is given for every group or for none
  if (rmask1 != nullptr)
    for (int i = 0; i < ngroups; ++i) uniform = uniform && ((rmask1[i] != nullptr) == (rmask1[0] != nullptr));
  SERL_CHECK_ARG(uniform, "convs_dgrad_wgrad1: the ReLU mask must be given for every group or for none");
  PConvDgw1Args a;
  a.B = B; a.IH1 = IH1; a.rb1 = IW1 * C1; a.scale = 1.0f / 255.0f;
  a.IH = (IH1 - 3) / 2 + 1; a.IW = (IW1 - 3) / 2 + 1;
  a.OH = (a.IH - 3) / 2 + 1; a.OW = (a.IW - 3) / 2 + 1;
  a.nbands = cdiv((a.IH + 1) / 2, PDgw1Cfg::P);
  a.d_bands = make_fastdiv(a.nbands);
  a.slab = workspace;
  for (int i = 0; i < ngroups; ++i) {
    SERL_CHECK_ARG(dw1[i] && dbias1[i], "convs_dgrad_wgrad1: null gradient output");
    a.dz[i] = (const ushort_t*)dz2[i]; a.wd[i] = (const ushort_t*)wd2[i];
    a.xbits[i] = rmask1 ? (const unsigned char*)rmask1[i] : nullptr;
    a.img[i] = (const unsigned char*)frames_u8[i];
  }
  int nslab = 0;
  if (int e = pconv_dgw1_launch(a, ngroups, workspace_bytes, &nslab, (hipStream_t)stream, "convs_dgrad_wgrad1")) return e;
  SERL_CHECK_ARG(nslab > 0, "convs_dgrad_wgrad1: the workspace (%lld bytes) cannot hold the slabs (serl_convs_dgrad_wgrad1_workspace_bytes)",
                 workspace_bytes);
  ReduceArgs ra; ra.slab = workspace; ra.nsplit = nslab; ra.kkc = 9 * C1; ra.F = F1;
  for (int i = 0; i < ngroups; ++i) { ra.dw[i] = dw1[i]; ra.db[i] = dbias1[i]; }
  launch_wgrad_reduce(ra, ngroups, (hipStream_t)stream);
  SERL_CHECK_LAUNCH("wgrad_reduce");
  return SERL_OK;
}

static int convs_wgrad_impl(const char* name, int ngroups, const void* const* x, const void* const* dz,
                            float* const* dw, float* const* dbias, float* workspace, long long workspace_bytes,
                            int in_u8, int B, int IH, int IW, int C, int KH, int KW, int stride, int F, void* stream) {
  SERL_CHECK_ARG(ngroups >= 1 && ngroups <= MAXG, "%s: ngroups must be in [1,%d]", name, MAXG);
  ConvGeom g;
  if (int e = make_geom(g, B, IH, IW, C, KH, KW, stride, F)) return e;
  SERL_CHECK_ARG(F % 8 == 0 && (in_u8 ? (KW * C >= 8 && KW * C <= 32) : C % 8 == 0), "%s: need F %% 8 == 0 and C %% 8 == 0 (uint8 input: 8 <= KW*C <= 32)", name);
  SERL_CHECK_ARG(all_aligned16(dz, ngroups) && (in_u8 || all_aligned16(x, ngroups)), "%s: S-format tensors must be 16-byte aligned", name);
  int nsplit, klen;
  wgrad_plan(g, ngroups, nsplit, klen);
  const int kkc = KH * KW * C;
  if (g_patch_kernels && !in_u8 && KH == 3 && KW == 3 && stride == 2 && C == 32 && F == 64 && g.OW <= 31 &&
      (IW + 1) / 2 <= 32 && workspace != nullptr) {
    // x band + dz band staged once in LDS, operands read with the transposing LDS load (pconv.h); slabs as below
    PConvWgradArgs a;
    a.B = B; a.IH = IH; a.IW = IW; a.OH = g.OH; a.OW = g.OW; a.slab = workspace;
    a.nbands = cdiv(g.OH, PWgradCfg::R);
    a.d_bands = make_fastdiv(a.nbands);
    for (int i = 0; i < ngroups; ++i) { a.x[i] = (const ushort_t*)x[i]; a.dz[i] = (const ushort_t*)dz[i]; }
    int nslab = 0;
    if (int e = pconv_wgrad_launch(a, ngroups, nsplit, workspace_bytes, &nslab, (hipStream_t)stream, name)) return e;
    if (nslab > 0) {
      ReduceArgs ra; ra.slab = workspace; ra.nsplit = nslab; ra.kkc = kkc; ra.F = F;
      for (int i = 0; i < ngroups; ++i) { ra.dw[i] = dw[i]; ra.db[i] = dbias[i]; }
      launch_wgrad_reduce(ra, ngroups, (hipStream_t)stream);
      SERL_CHECK_LAUNCH("wgrad_reduce");
      return SERL_OK;
    }
  }
  if (g_patch_kernels && g_wide_kernels && g_wgradw && !in_u8 && KH == 3 && KW == 3 && stride == 2 && C == 64 && F == 128 &&
      g.OW <= 15 && (IW + 1) / 2 <= 16 && workspace != nullptr) {
    // the same scheme for the third layer (pconv_wgradw_kernel): 76 accumulator tiles persistent in an 8-wave block
    PConvWgradArgs a;
    a.B = B; a.IH = IH; a.IW = IW; a.OH = g.OH; a.OW = g.OW; a.slab = workspace;
    a.nbands = cdiv(g.OH, PWgradWCfg::R);
    a.d_bands = make_fastdiv(a.nbands);
    for (int i = 0; i < ngroups; ++i) { a.x[i] = (const ushort_t*)x[i]; a.dz[i] = (const ushort_t*)dz[i]; }
    int nslab = 0;
    if (int e = pconv_wgradw_launch(a, ngroups, nsplit, workspace_bytes, &nslab, (hipStream_t)stream, name)) return e;
    if (nslab > 0) {
      ReduceArgs ra; ra.slab = workspace; ra.nsplit = nslab; ra.kkc = kkc; ra.F = F;
      for (int i = 0; i < ngroups; ++i) { ra.dw[i] = dw[i]; ra.db[i] = dbias[i]; }
      launch_wgrad_reduce(ra, ngroups, (hipStream_t)stream);
      SERL_CHECK_LAUNCH("wgrad_reduce");
      return SERL_OK;
    }
  }
  const long long need = (long long)ngroups * nsplit * (kkc + 1) * F * 4ll;
  SERL_CHECK_ARG(workspace != nullptr && workspace_bytes >= need, "%s: split-K slabs need %lld bytes, %lld available",
                 name, need, workspace_bytes);
  hipStream_t st = (hipStream_t)stream;
  int e;
#define FILLW(p)                                                                                      \
  p.g = g; p.slab = workspace; p.nsplit = nsplit; p.klen = klen; p.kkc = kkc;                          \
  p.Mp = in_u8 ? KH * 32 + 1 : kkc + 1; p.scale = in_u8 ? 1.0f / 255.0f : 1.0f;                        \
  p.d_c = make_fastdiv(C); p.d_kw = make_fastdiv(KW);                                                  \
  for (int i = 0; i < ngroups; ++i) { p.x[i] = x[i]; p.dz[i] = (const ushort_t*)dz[i]; }
  if (in_u8) { ConvWgradS<true> p; FILLW(p) e = s_launch_by_n(S_WGRAD, p, p.Mp, F, nsplit, ngroups, st, "convs_wgrad_u8"); }
  else { ConvWgradS<false> p; FILLW(p) e = s_launch_by_n(S_WGRAD, p, p.Mp, F, nsplit, ngroups, st, "convs_wgrad"); }
#undef FILLW
  if (e) return e;
  ReduceArgs ra; ra.slab = workspace; ra.nsplit = nsplit; ra.kkc = kkc; ra.F = F;
  for (int i = 0; i < ngroups; ++i) { ra.dw[i] = dw[i]; ra.db[i] = dbias[i]; }
  launch_wgrad_reduce(ra, ngroups, st);
  SERL_CHECK_LAUNCH("wgrad_reduce");
  return SERL_OK;
}

extern "C" int serl_convs_wgrad(int ngroups, const void* const* x, const void* const* dz, float* const* dw,
                                float* const* dbias, float* workspace, long long workspace_bytes, int in_u8, int B,
                                int IH, int IW, int C, int KH, int KW, int stride, int F, void* stream) {
  return convs_wgrad_impl("convs_wgrad", ngroups, x, dz, dw, dbias, workspace, workspace_bytes, in_u8, B, IH, IW, C, KH,
                          KW, stride, F, stream);
}

// ---- the same weight gradient for callers that cannot allocate (HIP graph capture, serving loops, harnesses that
// forbid hipMalloc): the split-K slabs live in a file-scope device array instead of a caller-provided workspace.
// Every GPU the library runs on holds its own 64 MiB copy of the array until the process ends.  In host code the
// array's name is only a stand-in: the device address is looked up once per device with hipGetSymbolAddress and
// that pointer is what the kernels get.  The slabs are written completely by the first kernel before the reduction
// reads them, so nothing has to be cleared between calls; calls that overlap on one GPU (two streams) race on it.
#define SERL_WGRAD_SCRATCH_BYTES (64ll << 20)
#define SERL_MAX_DEVICES 64
static __device__ __align__(256) float scratch[SERL_WGRAD_SCRATCH_BYTES / 4];
static float* g_scratch_ptr[SERL_MAX_DEVICES] = {};
static std::mutex g_scratch_mu;

static int wgrad_scratch_ptr(float** out) {
  int dev = -1;
  SERL_CHECK_HIP(hipGetDevice(&dev), "convs_wgrad_noalloc");
  SERL_CHECK_ARG(dev >= 0 && dev < SERL_MAX_DEVICES, "convs_wgrad_noalloc: device ordinal %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (g_scratch_ptr[dev] == nullptr) {         // set-up, once per device, with that device current
    void* ptr = nullptr;
    SERL_CHECK_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)), "convs_wgrad_noalloc");
    SERL_CHECK_ARG(ptr != nullptr, "convs_wgrad_noalloc: hipGetSymbolAddress returned NULL");
    g_scratch_ptr[dev] = (float*)ptr;
  }
  *out = g_scratch_ptr[dev];
  return SERL_OK;
}

extern "C" long long serl_convs_wgrad_noalloc_max_bytes(void) { return SERL_WGRAD_SCRATCH_BYTES; }

extern "C" int serl_convs_wgrad_noalloc(int ngroups, const void* const* x, const void* const* dz, float* const* dw,
                                        float* const* dbias, int in_u8, int B, int IH, int IW, int C, int KH, int KW,
                                        int stride, int F, void* stream) {
  float* ptr = nullptr;
  if (int e = wgrad_scratch_ptr(&ptr)) return e;
  // rejects shapes whose slabs exceed the array ("... need N bytes, 67108864 available")
  return convs_wgrad_impl("convs_wgrad_noalloc", ngroups, x, dz, dw, dbias, ptr, SERL_WGRAD_SCRATCH_BYTES, in_u8, B, IH,
                          IW, C, KH, KW, stride, F, stream);
}

#ifdef PFPP_TRACE
extern "C" int serl_debug_pf_trace(unsigned int* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pf_trace), sizeof(unsigned int) * 256) == hipSuccess ? 0 : -1;
}
#endif
