// Batched / segmented dense GEMM on the exact-fp32 MFMA pipe: every matmul of the MLP trunk, the
// critic ensemble (flax nn.vmap over a leading ensemble axis, serl/networks/ensemble.py:14-22) and
// the policy head, forward and backward, is one call of serl_gemm with different strides.
//
//   C[z][m][n] (+)= sum_k A[z][m][k] * B[z][k][n]          z = blockIdx.z (ensemble member)
//   A elem = a[z*a_zs + seg*a_ss + m*a_ms + r*a_ks],  B elem = b[z*b_zs + seg*b_ss + r*b_ks + n*b_ns]
//   with k = seg*seglen + r  (segments let one GEMM reduce over (member, feature) pairs).
#include <stdlib.h>
#include "igemm_bf3.h"

#ifndef GEMM_DEPTH
#define GEMM_DEPTH 1   // K-steps of register prefetch of the small-tile bf16_3x kernel (igemm_bf3.h); 2 measured neutral, 3 spills
#endif

struct GemmArgs {
  const float* a; const float* b; float* c; const float* bias;
  int M, N, K, seglen;
  long long a_zs, a_ss, a_ms, a_ks;
  long long b_zs, b_ss, b_ks, b_ns;
  long long c_zs, c_ss /*split stride*/, c_ms;
  long long bias_zs;
  int klen, relu, accumulate;
  FastDiv d_seg;
};

template <bool AK_, bool BN_, int VA_, int VB_>
struct GemmP {
  static constexpr bool A_KCONTIG = AK_, B_NCONTIG = BN_;
  static constexpr int VA = VA_, VB = VB_;
  GemmArgs g;
  struct Ctx { const float* a; const float* b; float* c; const float* bias; };
  struct ARow { long long off; bool ok; };
  struct AK { long long off; bool ok; };
  struct BCol { long long off; bool ok; };
  struct BK_ { long long off; bool ok; };
  struct ORow { long long off; };
  struct OCol { float bias; };
  __device__ bool init(Ctx& c, int z) const {
    c.a = g.a + z * g.a_zs; c.b = g.b + z * g.b_zs; c.c = g.c + z * g.c_zs;
    c.bias = g.bias ? g.bias + z * g.bias_zs : nullptr;
    return true;
  }
  __device__ int M(const Ctx&) const { return g.M; }
  __device__ int N(const Ctx&) const { return g.N; }
  __device__ int K(const Ctx&) const { return g.K; }
  __device__ int ksplit_len() const { return g.klen; }
  __device__ void a_row(const Ctx&, int m, ARow& r) const { r.ok = m < g.M; r.off = r.ok ? m * g.a_ms : 0; }
  __device__ void a_k(const Ctx&, int k, int kend, AK& a) const {
    a.ok = k < kend;
    const uint32_t kk = a.ok ? k : 0;
    const uint32_t seg = fd_div(kk, g.d_seg), r = kk - seg * g.seglen;
    a.off = seg * g.a_ss + r * g.a_ks;
  }
  __device__ void a_load(const Ctx& c, const ARow& r, const AK& a, VecF<VA_>& raw, float& mul, VecF<VA_>& add) const {
    const bool ok = r.ok && a.ok;
    mul = ok ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < VA_; ++j) add.v[j] = 0.f;
    ld_vec<VA_>(c.a + (ok ? r.off + a.off : 0), raw);
  }
  __device__ void b_col(const Ctx&, int n, BCol& b) const { b.ok = n < g.N; b.off = b.ok ? n * g.b_ns : 0; }
  __device__ void b_k(const Ctx&, int k, int kend, BK_& b) const {
    b.ok = k < kend;
    const uint32_t kk = b.ok ? k : 0;
    const uint32_t seg = fd_div(kk, g.d_seg), r = kk - seg * g.seglen;
    b.off = seg * g.b_ss + r * g.b_ks;
  }
  __device__ void b_load(const Ctx& c, const BK_& k, const BCol& n, VecF<VB_>& raw, float& mul) const {
    const bool ok = k.ok && n.ok;
    mul = ok ? 1.f : 0.f;
    ld_vec<VB_>(c.b + (ok ? k.off + n.off : 0), raw);
  }
  __device__ void o_row(const Ctx&, int m, ORow& o) const { o.off = m * g.c_ms; }
  __device__ void o_col(const Ctx& c, int n, OCol& o) const { o.bias = c.bias ? c.bias[n] : 0.f; }
  __device__ float o_aux(const Ctx& c, const ORow& o, int n, int split) const {
    return g.accumulate ? c.c[n >= 0 ? split * g.c_ss + o.off + n : 0] : 0.f;
  }
  __device__ void store(const Ctx& c, const ORow& o, const OCol& oc, int n, int split, float v, float aux) const {
    v += oc.bias + aux;
    if (g.relu == 1) v = fmaxf(v, 0.f);                                              // activation code: 1 ReLU,
    else if (g.relu == 2) v = v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);     // 2 hard-swish,
    else if (g.relu == 3) v = fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);         // 3 hard-sigmoid (MobileNetV3),
    else if (g.relu == 4) v = 1.f / (1.f + expf(-v));                                // 4 sigmoid (reward classifiers)
    c.c[split * g.c_ss + o.off + n] = v;
  }
};

typedef TileCfg<64, 64, 16, 2, 2> T64x64;
typedef TileCfg<128, 128, 16, 2, 2> T128x128;
// bf16_3x: the learner's dense GEMMs are 16..160 blocks of a long K loop, i.e. bound by the latency of one
// K-step (~1 us: global -> LDS -> MFMA); a 64-deep slab makes 4x fewer of them.
typedef TileCfg<64, 64, 64, 2, 2> B64x64;
typedef TileCfg<128, 128, 32, 2, 2> B128x128;

// layout codes: a_kcontig = 1 when consecutive k of A are adjacent in memory (else consecutive m are),
//               b_ncontig = 1 when consecutive n of B are adjacent (else consecutive k are).
static thread_local int g_gemm_big_n = 0;
void serl_gemm_set_big_n(int n) { g_gemm_big_n = n; }

extern "C" int serl_gemm(const float* a, const float* b, float* c, const float* bias, int nbatch, int M, int N, int K,
                         int seglen, long long a_zs, long long a_ss, long long a_ms, long long a_ks, long long b_zs,
                         long long b_ss, long long b_ks, long long b_ns, long long c_zs, long long c_ms,
                         long long bias_zs, int nsplit, long long c_split_stride, int relu, int accumulate,
                         int a_kcontig, int b_ncontig, void* stream) {
  SERL_CHECK_ARG(a && b && c, "gemm: null pointer");
  SERL_CHECK_ARG(nbatch >= 1 && M >= 1 && N >= 1 && K >= 1 && nsplit >= 1, "gemm: bad sizes");
  SERL_CHECK_ARG(seglen >= 1 && K % seglen == 0, "gemm: K must be a multiple of seglen");
  SERL_CHECK_ARG(nsplit == 1 || (!bias && !relu && !accumulate), "gemm: split-K output is raw partial sums");
  GemmArgs g;
  g.a = a; g.b = b; g.c = c; g.bias = bias; g.M = M; g.N = N; g.K = K; g.seglen = seglen;
  g.a_zs = a_zs; g.a_ss = a_ss; g.a_ms = a_ms; g.a_ks = a_ks;
  g.b_zs = b_zs; g.b_ss = b_ss; g.b_ks = b_ks; g.b_ns = b_ns;
  g.c_zs = c_zs; g.c_ss = c_split_stride; g.c_ms = c_ms; g.bias_zs = bias_zs;
  g.relu = relu; g.accumulate = accumulate;
  g.d_seg = make_fastdiv(seglen);
  int klen = cdiv(cdiv(K, nsplit), 16) * 16;   // (slabs of the bf16_3x kernel may end inside a K-step: masked)
  g.klen = klen;
  const int ns = cdiv(K, klen);
  SERL_CHECK_ARG(ns == nsplit, "gemm: nsplit %d does not tile K=%d in multiples of 16 (got %d)", nsplit, K, ns);
  hipStream_t st = (hipStream_t)stream;
  // 128 x 128 tiles from this many columns on (and 512 rows): 512 for the learner's dense layers; the frozen MobileNetV3's
  // pointwise convs (16 k .. 1 M rows x 80 .. 960 columns, memory-bound: fewer re-reads and re-splits of the activation
  // operand per column tile) lower it for their calls (serl_gemm_set_big_n, per calling thread).  Same bits either way:
  // every tile walks k in the same order.
  const bool big = (M >= 512 && N >= (g_gemm_big_n > 0 ? g_gemm_big_n : 512));
  // float4 loads along the contiguous axis when every vector of 4 is aligned, contiguous and all-or-nothing valid
  auto m4 = [](long long v) { return (v & 3) == 0; };
  const bool a4 = (((uintptr_t)a) & 15) == 0 && m4(a_zs) && m4(a_ss) &&
                  (a_kcontig ? (a_ks == 1 && m4(seglen) && m4(a_ms)) : (a_ms == 1 && m4(M) && m4(a_ks)));
  const bool b4 = (((uintptr_t)b) & 15) == 0 && m4(b_zs) && m4(b_ss) &&
                  (b_ncontig ? (b_ns == 1 && m4(N) && m4(b_ks)) : (b_ks == 1 && m4(seglen) && m4(b_ns)));
#define GO2(AK, BN, VA, VB)                                                                                \
  {                                                                                                         \
    GemmP<AK, BN, VA, VB> p; p.g = g;                                                                       \
    return big ? igemm_launch_prec<GemmP<AK, BN, VA, VB>, T128x128, B128x128>(p, M, N, nsplit, nbatch, st, "gemm") \
               : igemm_launch_prec<GemmP<AK, BN, VA, VB>, T64x64, B64x64, GEMM_DEPTH>(p, M, N, nsplit, nbatch, st, "gemm");  \
  }
#define GO(AK, BN)                     \
  {                                    \
    if (a4 && b4) GO2(AK, BN, 4, 4)    \
    if (a4) GO2(AK, BN, 4, 1)          \
    if (b4) GO2(AK, BN, 1, 4)          \
    GO2(AK, BN, 1, 1)                  \
  }
  if (a_kcontig && b_ncontig) GO(true, true)
  if (a_kcontig && !b_ncontig) GO(true, false)
  if (!a_kcontig && b_ncontig) GO(false, true)
  GO(false, false)
#undef GO
#undef GO2
}
