// Row-wise and elementwise kernels of the SAC / RLPD update (HBM- or latency-bound; wave64
// shuffles for the row statistics, float4 streams for the optimiser).
#include "common.h"

#define LN_EPS 1e-6f  // flax.linen.LayerNorm default (serl/networks/mlp.py:33)
#define ACT_NONE 0
#define ACT_RELU 1
#define ACT_TANH 2
#define ACT_LEAKY 3   // jax.nn.leaky_relu, negative_slope 0.01 (VICE classifier MLP, vice_learner.py:156)
#define MAXW 16  // row width <= 64*MAXW = 1024

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ------------------------------------------------------------------ bias + LayerNorm + activation
// z = sum_s pre[s] + bias ; y = act(LN(z)) (or act(z) when use_ln == 0).  One wave per row.
// Rows are grouped (ensemble member e = row / rows_per_group) for bias / scale / shift.
// flax 0.7 LayerNorm: var = max(0, E[z^2]-E[z]^2); y = (z-mean) * (rsqrt(var+eps)*scale) + shift.
struct LnFwdArgs {
  const float* pre; long long pre_ss; int nslab; int ldpre;
  const float* bias; const float* gamma; const float* beta; long long pgs;  // per-group stride of params
  float* y; int ldy; float* xhat; float* rstd;
  int R, N, rows_per_group, use_ln, act;
  // optional Dense(1) head on the activated row (state_action_value.py:17-19): q[row] = y[row,:] . hw[e,:] + hb[e]
  const float* hw; long long hw_gs; const float* hb; long long hb_gs; float* q;
};
__global__ __launch_bounds__(256) void ln_act_fwd_kernel(const LnFwdArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= a.R) return;
  const int e = row / a.rows_per_group;
  const float* bias = a.bias ? a.bias + e * a.pgs : nullptr;
  float z[MAXW];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int n = lane + 64 * i;
    float v = 0.f;
    if (n < a.N) {
      const float* src = a.pre + (long long)row * a.ldpre + n;
#pragma unroll 8   // the slab loads are independent: issue them together (a rolled loop pays one round trip each)
      for (int s = 0; s < a.nslab; ++s) v += src[s * a.pre_ss];
      if (bias) v += bias[n];
    }
    z[i] = v;
    s1 += v;
    s2 = fmaf(v, v, s2);          // explicit FMAs: ensemble.hip repeats this arithmetic and must round identically
  }
  float mean = 0.f, mul = 1.f;
  if (a.use_ln) {
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    mean = s1 / (float)a.N;
    const float var = fmaxf(fmaf(-mean, mean, s2 / (float)a.N), 0.f);
    mul = rsqrtf(var + LN_EPS);
    if (a.rstd && lane == 0) a.rstd[row] = mul;
  }
  const float* gamma = a.use_ln ? a.gamma + e * a.pgs : nullptr;
  const float* beta = a.use_ln ? a.beta + e * a.pgs : nullptr;
  const float* hw = a.hw ? a.hw + e * a.hw_gs : nullptr;
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int n = lane + 64 * i;
    if (n < a.N) {
      float v = z[i];
      if (a.use_ln) {
        const float xh = (v - mean) * mul;
        if (a.xhat) a.xhat[(long long)row * a.N + n] = xh;
        v = fmaf(v - mean, mul * gamma[n], beta[n]);
      }
      if (a.act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (a.act == ACT_TANH) v = tanhf(v);
      else if (a.act == ACT_LEAKY) v = v > 0.f ? v : 0.01f * v;
      a.y[(long long)row * a.ldy + n] = v;
      if (hw) dot = fmaf(v, hw[n], dot);                 // same lane/column order as rowdot_kernel: bit-identical sums
    }
  }
  if (hw) {
    dot = wave_sum(dot);
    if (lane == 0) a.q[row] = dot + a.hb[e * a.hb_gs];
  }
}

extern "C" int serl_ln_act_fwd(const float* pre, int nslab, long long pre_slab_stride, int ldpre, const float* bias,
                               const float* gamma, const float* beta, long long param_group_stride, float* y, int ldy,
                               float* xhat, float* rstd, int R, int N, int rows_per_group, int use_ln, int act,
                               void* stream) {
  SERL_CHECK_ARG(pre && y && R >= 1 && N >= 1 && N <= 64 * MAXW, "ln_act_fwd: bad args (N must be <= %d)", 64 * MAXW);
  SERL_CHECK_ARG(!use_ln || (gamma && beta), "ln_act_fwd: LayerNorm needs scale and bias");
  SERL_CHECK_ARG(rows_per_group >= 1 && nslab >= 1, "ln_act_fwd: bad grouping");
  LnFwdArgs a{pre, pre_slab_stride, nslab, ldpre, bias, gamma, beta, param_group_stride, y, ldy, xhat, rstd,
              R, N, rows_per_group, use_ln, act, nullptr, 0, nullptr, 0, nullptr};
  hipLaunchKernelGGL(ln_act_fwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, a);
  SERL_CHECK_LAUNCH("ln_act_fwd");
  return SERL_OK;
}

// the same with the Dense(1) Q head of the ensemble fused behind the activation (one launch less per ensemble pass)
extern "C" int serl_ln_act_head_fwd(const float* pre, int nslab, long long pre_slab_stride, int ldpre, const float* bias,
                                    const float* gamma, const float* beta, long long param_group_stride, float* y,
                                    int ldy, float* xhat, float* rstd, int R, int N, int rows_per_group, int use_ln,
                                    int act, const float* head_w, long long head_w_group_stride, const float* head_b,
                                    long long head_b_group_stride, float* q, void* stream) {
  SERL_CHECK_ARG(pre && y && R >= 1 && N >= 1 && N <= 64 * MAXW, "ln_act_head_fwd: bad args (N must be <= %d)", 64 * MAXW);
  SERL_CHECK_ARG(!use_ln || (gamma && beta), "ln_act_head_fwd: LayerNorm needs scale and bias");
  SERL_CHECK_ARG(rows_per_group >= 1 && nslab >= 1, "ln_act_head_fwd: bad grouping");
  SERL_CHECK_ARG(head_w && head_b && q, "ln_act_head_fwd: head kernel, bias and output are required");
  LnFwdArgs a{pre, pre_slab_stride, nslab, ldpre, bias, gamma, beta, param_group_stride, y, ldy, xhat, rstd,
              R, N, rows_per_group, use_ln, act, head_w, head_w_group_stride, head_b, head_b_group_stride, q};
  hipLaunchKernelGGL(ln_act_fwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, a);
  SERL_CHECK_LAUNCH("ln_act_head_fwd");
  return SERL_OK;
}

// PixelMultiplexer trunk (serl/networks/pixel_multiplexer.py:34-43) behind the split-K GEMM of Dense_0, one launch:
//   blockIdx.y = 0:  y0 = act(LN(sum of nslab partials + bias0))                      (image branch, N0 columns)
//   blockIdx.y = 1:  y1 = act(LN(x1[row, :K1] . W1[K1][N1] + bias1))                  (state branch: K1 ~ 19, fp32 FMA)
struct TrunkLnArgs {
  const float* slab; long long slab_ss; int nslab; int N0;
  const float* bias0; const float* gamma0; const float* beta0; float* y0; float* xhat0; float* rstd0;
  const float* x1; int ldx1; int K1; const float* W1; int N1;
  const float* bias1; const float* gamma1; const float* beta1; float* y1; float* xhat1; float* rstd1;
  int ldy, R, act;
};
__global__ __launch_bounds__(256) void trunk_ln_fwd_kernel(const TrunkLnArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= a.R) return;
  const bool img = blockIdx.y == 0;
  const int N = img ? a.N0 : a.N1;
  const float* bias = img ? a.bias0 : a.bias1;
  const float* gamma = img ? a.gamma0 : a.gamma1;
  const float* beta = img ? a.beta0 : a.beta1;
  float* y = img ? a.y0 : a.y1;
  float* xhat = img ? a.xhat0 : a.xhat1;
  float* rstd = img ? a.rstd0 : a.rstd1;
  float z[MAXW];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int n = lane + 64 * i;
    float v = 0.f;
    if (n < N) {
      if (img) {
        const float* src = a.slab + (long long)row * a.N0 + n;
#pragma unroll 8
        for (int s = 0; s < a.nslab; ++s) v += src[s * a.slab_ss];
      } else {
        const float* xr = a.x1 + (long long)row * a.ldx1;
        for (int k = 0; k < a.K1; ++k) v = fmaf(xr[k], a.W1[k * a.N1 + n], v);
      }
      v += bias[n];
    }
    z[i] = v;
    s1 += v;
    s2 += v * v;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const float mean = s1 / (float)N;
  const float mul = rsqrtf(fmaxf(s2 / (float)N - mean * mean, 0.f) + LN_EPS);
  if (rstd && lane == 0) rstd[row] = mul;
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int n = lane + 64 * i;
    if (n < N) {
      const float xh = (z[i] - mean) * mul;
      if (xhat) xhat[(long long)row * N + n] = xh;
      float v = (z[i] - mean) * (mul * gamma[n]) + beta[n];
      if (a.act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (a.act == ACT_TANH) v = tanhf(v);
      else if (a.act == ACT_LEAKY) v = v > 0.f ? v : 0.01f * v;
      y[(long long)row * a.ldy + n] = v;
    }
  }
}
extern "C" int serl_trunk_ln_fwd(const float* slab, int nslab, long long slab_stride, int N0, const float* bias0,
                                 const float* gamma0, const float* beta0, float* y0, float* xhat0, float* rstd0,
                                 const float* x1, int ldx1, int K1, const float* W1, int N1, const float* bias1,
                                 const float* gamma1, const float* beta1, float* y1, float* xhat1, float* rstd1,
                                 int ldy, int R, int act, void* stream) {
  SERL_CHECK_ARG(slab && bias0 && gamma0 && beta0 && y0 && x1 && W1 && bias1 && gamma1 && beta1 && y1, "trunk_ln_fwd: null pointer");
  SERL_CHECK_ARG(nslab >= 1 && R >= 1 && K1 >= 1 && N0 >= 1 && N0 <= 64 * MAXW && N1 >= 1 && N1 <= 64 * MAXW,
                 "trunk_ln_fwd: bad sizes (N0, N1 must be <= %d)", 64 * MAXW);
  TrunkLnArgs a{slab, slab_stride, nslab, N0, bias0, gamma0, beta0, y0, xhat0, rstd0, x1, ldx1, K1, W1, N1,
                bias1, gamma1, beta1, y1, xhat1, rstd1, ldy, R, act};
  hipLaunchKernelGGL(trunk_ln_fwd_kernel, dim3(cdiv(R, 4), 2), dim3(256), 0, (hipStream_t)stream, a);
  SERL_CHECK_LAUNCH("trunk_ln_fwd");
  return SERL_OK;
}

// Backward of act(LN(z)): g = dy * act'(y); dxhat = g*gamma;
// dz = rstd * (dxhat - mean(dxhat) - xhat*mean(dxhat*xhat)).  dy is either a dense (R,N) matrix with
// row stride lddy, or the rank-1 form dq[row] * w[e][n] (gradient through a Dense(1) head).
struct LnBwdArgs {
  const float* dy; int lddy; const float* dq; const float* w_head; long long w_gs;
  const float* y; int ldy; const float* xhat; const float* rstd; const float* gamma; long long pgs;
  float* dz; float* g;
  int R, N, rows_per_group, use_ln, act;
};
__global__ __launch_bounds__(256) void ln_act_bwd_kernel(const LnBwdArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= a.R) return;
  const int e = row / a.rows_per_group;
  const float* gamma = a.use_ln ? a.gamma + e * a.pgs : nullptr;
  const float* wh = a.dq ? a.w_head + e * a.w_gs : nullptr;
  const float dq = a.dq ? a.dq[row] : 0.f;
  float gv[MAXW], xh[MAXW];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int n = lane + 64 * i;
    float g = 0.f, x = 0.f;
    if (n < a.N) {
      const float d = a.dq ? dq * wh[n] : a.dy[(long long)row * a.lddy + n];
      const float yv = a.y[(long long)row * a.ldy + n];
      float dact = 1.f;
      if (a.act == ACT_RELU) dact = yv > 0.f ? 1.f : 0.f;
      else if (a.act == ACT_TANH) dact = 1.f - yv * yv;
      else if (a.act == ACT_LEAKY) dact = yv > 0.f ? 1.f : 0.01f;
      g = d * dact;
      if (a.use_ln) {
        x = a.xhat[(long long)row * a.N + n];
        const float dxh = g * gamma[n];
        s1 += dxh;
        s2 += dxh * x;
      }
    }
    gv[i] = g;
    xh[i] = x;
  }
  float m1 = 0.f, m2 = 0.f, rs = 1.f;
  if (a.use_ln) {
    m1 = wave_sum(s1) / (float)a.N;
    m2 = wave_sum(s2) / (float)a.N;
    rs = a.rstd[row];
  }
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int n = lane + 64 * i;
    if (n < a.N) {
      float dz = gv[i];
      if (a.use_ln) {
        dz = rs * (gv[i] * gamma[n] - m1 - xh[i] * m2);
        if (a.g) a.g[(long long)row * a.N + n] = gv[i];
      }
      a.dz[(long long)row * a.N + n] = dz;
    }
  }
}

extern "C" int serl_ln_act_bwd(const float* dy, int lddy, const float* dq, const float* w_head, long long w_group_stride,
                               const float* y, int ldy, const float* xhat, const float* rstd, const float* gamma,
                               long long param_group_stride, float* dz, float* g, int R, int N, int rows_per_group,
                               int use_ln, int act, void* stream) {
  SERL_CHECK_ARG((dy != nullptr) != (dq != nullptr), "ln_act_bwd: give exactly one of dy / dq");
  SERL_CHECK_ARG(!dq || w_head, "ln_act_bwd: rank-1 form needs the head weights");
  SERL_CHECK_ARG(y && dz && R >= 1 && N >= 1 && N <= 64 * MAXW, "ln_act_bwd: bad args");
  SERL_CHECK_ARG(!use_ln || (xhat && rstd && gamma && g), "ln_act_bwd: LayerNorm needs xhat, rstd, scale, g");
  LnBwdArgs a{dy, lddy, dq, w_head, w_group_stride, y, ldy, xhat, rstd, gamma, param_group_stride, dz, g,
              R, N, rows_per_group, use_ln, act};
  hipLaunchKernelGGL(ln_act_bwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, a);
  SERL_CHECK_LAUNCH("ln_act_bwd");
  return SERL_OK;
}

// Several act(LN(.)) backward problems over the SAME rows in one launch (blockIdx.y = segment), with dy given as the
// sum of `nslab` split-K partials: the trunk's two LayerNorm branches read their column ranges of the ensemble's
// input gradient straight from the split-K slabs of the producing GEMM (one launch instead of slab-sum + 2 x LN bwd).
#define LN_MAXSEG 4
struct LnBwdSegArgs {
  const float* dy[LN_MAXSEG]; int lddy[LN_MAXSEG]; const float* y[LN_MAXSEG]; int ldy[LN_MAXSEG];
  const float* xhat[LN_MAXSEG]; const float* rstd[LN_MAXSEG]; const float* gamma[LN_MAXSEG];
  float* dz[LN_MAXSEG]; float* g[LN_MAXSEG]; int N[LN_MAXSEG];
  int nslab; long long dy_ss; int R, act;
};
// MW = 64-column chunks a lane holds.  Every load is unconditional (columns past N read column N - 1 and contribute
// nothing) and the slab loop is the OUTER one, so a lane has MW x 4 independent loads in flight instead of one chain of
// nslab dependent round trips per chunk (39 -> 12 us at 256 rows x (512 + 64) columns x 8 slabs); the sum over the slabs
// keeps its order.
template <int MW>
__device__ __forceinline__ void ln_act_bwd_seg_body(const LnBwdSegArgs& a, int sg, int row, int lane) {
  const int N = a.N[sg];
  const float* gamma = a.gamma[sg];
  const float* src = a.dy[sg] + (long long)row * a.lddy[sg];
  int nn[MW];
  float d[MW], yv[MW], xh[MW], gm[MW];
#pragma unroll
  for (int i = 0; i < MW; ++i) {
    nn[i] = min(lane + 64 * i, N - 1);
    d[i] = 0.f;
    yv[i] = a.y[sg][(long long)row * a.ldy[sg] + nn[i]];
    xh[i] = a.xhat[sg][(long long)row * N + nn[i]];
    gm[i] = gamma[nn[i]];
  }
#pragma unroll 4
  for (int k = 0; k < a.nslab; ++k) {
    const float* sk = src + k * a.dy_ss;
#pragma unroll
    for (int i = 0; i < MW; ++i) d[i] += sk[nn[i]];
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MW; ++i) {
    float dact = 1.f;
    if (a.act == ACT_RELU) dact = yv[i] > 0.f ? 1.f : 0.f;
    else if (a.act == ACT_TANH) dact = 1.f - yv[i] * yv[i];
    else if (a.act == ACT_LEAKY) dact = yv[i] > 0.f ? 1.f : 0.01f;
    d[i] = d[i] * dact;                       // g
    if (lane + 64 * i < N) {
      const float dxh = d[i] * gm[i];
      s1 += dxh;
      s2 += dxh * xh[i];
    }
  }
  const float m1 = wave_sum(s1) / (float)N, m2 = wave_sum(s2) / (float)N, rs = a.rstd[sg][row];
#pragma unroll
  for (int i = 0; i < MW; ++i) {
    const int n = lane + 64 * i;
    if (n < N) {
      a.g[sg][(long long)row * N + n] = d[i];
      a.dz[sg][(long long)row * N + n] = rs * (d[i] * gm[i] - m1 - xh[i] * m2);
    }
  }
}
__global__ __launch_bounds__(256) void ln_act_bwd_seg_kernel(const LnBwdSegArgs a) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= a.R) return;
  const int sg = blockIdx.y;
  const int N = a.N[sg];              // block-uniform
  if (N <= 64) ln_act_bwd_seg_body<1>(a, sg, row, lane);
  else if (N <= 256) ln_act_bwd_seg_body<4>(a, sg, row, lane);
  else if (N <= 512) ln_act_bwd_seg_body<8>(a, sg, row, lane);
  else ln_act_bwd_seg_body<MAXW>(a, sg, row, lane);
}
extern "C" int serl_ln_act_bwd_seg(int nseg, const float* const* dy, const int* lddy, int dy_nslab,
                                   long long dy_slab_stride, const float* const* y, const int* ldy,
                                   const float* const* xhat, const float* const* rstd, const float* const* gamma,
                                   float* const* dz, float* const* g, int R, const int* N, int act, void* stream) {
  SERL_CHECK_ARG(nseg >= 1 && nseg <= LN_MAXSEG, "ln_act_bwd_seg: nseg must be in [1,%d]", LN_MAXSEG);
  SERL_CHECK_ARG(dy && lddy && y && ldy && xhat && rstd && gamma && dz && g && N, "ln_act_bwd_seg: null argument table");
  SERL_CHECK_ARG(R >= 1 && dy_nslab >= 1, "ln_act_bwd_seg: bad sizes");
  LnBwdSegArgs a;
  for (int i = 0; i < nseg; ++i) {
    SERL_CHECK_ARG(dy[i] && y[i] && xhat[i] && rstd[i] && gamma[i] && dz[i] && g[i], "ln_act_bwd_seg: null pointer");
    SERL_CHECK_ARG(N[i] >= 1 && N[i] <= 64 * MAXW, "ln_act_bwd_seg: N must be in [1,%d]", 64 * MAXW);
    a.dy[i] = dy[i]; a.lddy[i] = lddy[i]; a.y[i] = y[i]; a.ldy[i] = ldy[i]; a.xhat[i] = xhat[i]; a.rstd[i] = rstd[i];
    a.gamma[i] = gamma[i]; a.dz[i] = dz[i]; a.g[i] = g[i]; a.N[i] = N[i];
  }
  a.nslab = dy_nslab; a.dy_ss = dy_slab_stride; a.R = R; a.act = act;
  hipLaunchKernelGGL(ln_act_bwd_seg_kernel, dim3(cdiv(R, 4), nseg), dim3(256), 0, (hipStream_t)stream, a);
  SERL_CHECK_LAUNCH("ln_act_bwd_seg");
  return SERL_OK;
}

// ------------------------------------------------------------------ column reductions over the batch
// out_k[e][n] = sum_{r in group e} term_k[r][n], fixed summation order (reproducible):
//   k=0: x0[r][n] * (w ? w[r] : 1) * (mul ? mul[r][n] : 1)      k=1: x1[r][n]      k=2: x2[r][n] * mul2[r][n]
// Block = 16 columns x 16 row lanes; each row lane walks rows rl, rl+16, ... (unrolled: independent loads in flight)
struct ColSumArgs {
  const float* x0; const float* w; const float* mul; float* out0;
  const float* x1; float* out1;
  const float* x2; const float* mul2; float* out2;
  int rows_per_group, N, ldx; long long out_gs;
};
// The optional operands are template flags: a load behind a runtime condition makes hipcc branch around it and wait for
// every load separately, which turns the 16-row loop into 16 dependent memory round trips.
template <bool W, bool MUL, bool X1, bool X2>
__global__ __launch_bounds__(256) void colsum_kernel(const ColSumArgs a) {
  __shared__ float red[3][16][17];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int n = blockIdx.x * 16 + cl, e = blockIdx.y;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (n < a.N) {
#pragma unroll 8
    for (int r = rl; r < a.rows_per_group; r += 16) {
      const long long row = (long long)e * a.rows_per_group + r;
      const long long i = row * a.ldx + n;
      float v = a.x0[i];
      if constexpr (W) v *= a.w[row];
      if constexpr (MUL) v *= a.mul[i];
      s0 += v;
      if constexpr (X1) s1 += a.x1[i];
      if constexpr (X2) s2 += a.x2[i] * a.mul2[i];
    }
  }
  red[0][rl][cl] = s0; red[1][rl][cl] = s1; red[2][rl][cl] = s2;
  __syncthreads();
  if (rl < 3 && n < a.N) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += red[rl][k][cl];
    float* out = rl == 0 ? a.out0 : (rl == 1 ? a.out1 : a.out2);
    if (out) out[e * a.out_gs + n] = s;
  }
}

extern "C" int serl_colsum(const float* x0, const float* w, const float* mul, float* out0, const float* x1, float* out1,
                           const float* x2, const float* mul2, float* out2, int ngroups, int rows_per_group, int N,
                           int ldx, long long out_group_stride, void* stream) {
  SERL_CHECK_ARG(x0 && out0 && ngroups >= 1 && rows_per_group >= 1 && N >= 1, "colsum: bad args");
  SERL_CHECK_ARG((x1 != nullptr) == (out1 != nullptr), "colsum: x1/out1 mismatch");
  SERL_CHECK_ARG((x2 != nullptr) == (out2 != nullptr) && (!x2 || mul2), "colsum: x2/mul2/out2 mismatch");
  ColSumArgs a{x0, w, mul, out0, x1, out1, x2, mul2, out2, rows_per_group, N, ldx, out_group_stride};
  const dim3 grid(cdiv(N, 16), ngroups);
  hipStream_t st = (hipStream_t)stream;
#define COLSUM_CASE(W_, M_, X1_, X2_)                                                                                    \
  if ((w != nullptr) == W_ && (mul != nullptr) == M_ && (x1 != nullptr) == X1_ && (x2 != nullptr) == X2_)                \
    hipLaunchKernelGGL((colsum_kernel<W_, M_, X1_, X2_>), grid, dim3(256), 0, st, a);
  COLSUM_CASE(false, false, false, false) COLSUM_CASE(false, false, false, true) COLSUM_CASE(false, false, true, false)
  COLSUM_CASE(false, false, true, true) COLSUM_CASE(false, true, false, false) COLSUM_CASE(false, true, false, true)
  COLSUM_CASE(false, true, true, false) COLSUM_CASE(false, true, true, true) COLSUM_CASE(true, false, false, false)
  COLSUM_CASE(true, false, false, true) COLSUM_CASE(true, false, true, false) COLSUM_CASE(true, false, true, true)
  COLSUM_CASE(true, true, false, false) COLSUM_CASE(true, true, false, true) COLSUM_CASE(true, true, true, false)
  COLSUM_CASE(true, true, true, true)
#undef COLSUM_CASE
  SERL_CHECK_LAUNCH("colsum");
  return SERL_OK;
}

// ------------------------------------------------------------------ Dense(1) head: q[r] = h[r,:].w[e,:] + b[e]
__global__ __launch_bounds__(256) void rowdot_kernel(const float* h, const float* w, long long w_gs, const float* b,
                                                     long long b_gs, float* q, int R, int N, int rows_per_group) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  const int e = row / rows_per_group;
  float s = 0.f;
  for (int n = lane; n < N; n += 64) s = fmaf(h[(long long)row * N + n], w[e * w_gs + n], s);
  s = wave_sum(s);
  if (lane == 0) q[row] = s + b[e * b_gs];
}
extern "C" int serl_rowdot(const float* h, const float* w, long long w_group_stride, const float* b,
                           long long b_group_stride, float* q, int R, int N, int rows_per_group, void* stream) {
  SERL_CHECK_ARG(h && w && b && q && R >= 1 && N >= 1 && rows_per_group >= 1, "rowdot: bad args");
  hipLaunchKernelGGL(rowdot_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, h, w, w_group_stride, b,
                     b_group_stride, q, R, N, rows_per_group);
  SERL_CHECK_LAUNCH("rowdot");
  return SERL_OK;
}

// ------------------------------------------------------------------ strided 2-D copy / fill (concat plumbing)
__global__ void copy2d_kernel(float* dst, int ldd, const float* src, int lds, int rows, int cols) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
  dst[(long long)r * ldd + c] = src[(long long)r * lds + c];
}
extern "C" int serl_copy2d(float* dst, int ldd, const float* src, int lds, int rows, int cols, void* stream) {
  SERL_CHECK_ARG(dst && src && rows >= 1 && cols >= 1, "copy2d: bad args");
  const long long n = (long long)rows * cols;
  hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream, dst, ldd, src,
                     lds, rows, cols);
  SERL_CHECK_LAUNCH("copy2d");
  return SERL_OK;
}
__global__ void fill_kernel(float* p, float v, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
extern "C" int serl_fill(float* p, float v, long long n, void* stream) {
  SERL_CHECK_ARG(p && n >= 1, "fill: bad args");
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream, p, v, n);
  SERL_CHECK_LAUNCH("fill");
  return SERL_OK;
}

// ------------------------------------------------------------------ tanh-Gaussian policy head
// (erfinv_f32, jax_normal_from_bits, softplusf: common.h -- shared with the fused policy forward of ensemble.hip)

// One thread per row.  mean / log_std_raw: (B, A) with row stride ld_ms (the two Dense heads write
// side by side).  a = tanh(mean + exp(clip(log_std)) * eps); log pi(a) as TFP computes it from the
// cached pre-tanh sample (serl/distributions/tanh_normal.py:41-48).  eps index = global row * A + j
// of a (total_rows * A)-element jax.random.normal draw, so batch shards reproduce the global stream.
struct TgArgs {
  const float* mean; const float* lstd; int ld_ms;
  float* act; int ld_act; float* xpre; float* eps; float* logp;
  int B, A; uint32_t k0, k1; int row_offset, total_rows; int deterministic;
};
__global__ void tanh_gaussian_fwd_kernel(const TgArgs t) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= t.B) return;
  float lp = 0.f;
  for (int j = 0; j < t.A; ++j) {
    const float mu = t.mean[(long long)b * t.ld_ms + j];
    const float ls = fminf(fmaxf(t.lstd[(long long)b * t.ld_ms + j], -20.f), 2.f);
    float e = 0.f;
    if (!t.deterministic) {
      const uint32_t idx = (uint32_t)(t.row_offset + b) * t.A + j;
      e = jax_normal_from_bits(threefry_bits_at(t.k0, t.k1, idx, (uint32_t)t.total_rows * t.A));
    }
    const float x = mu + expf(ls) * e;
    t.act[(long long)b * t.ld_act + j] = tanhf(x);
    if (t.xpre) t.xpre[b * t.A + j] = x;
    if (t.eps) t.eps[b * t.A + j] = e;
    const float z = (x - mu) * expf(-ls);
    lp += (-0.5f * z * z - ls - 0.918938533204672742f) - 2.f * (0.693147180559945309f - x - softplusf(-2.f * x));
  }
  if (t.logp) t.logp[b] = lp;
}
extern "C" int serl_tanh_gaussian_fwd(const float* mean, const float* log_std_raw, int ld_ms, float* act, int ld_act,
                                      float* xpre, float* eps, float* logp, int B, int A, unsigned k0, unsigned k1,
                                      int row_offset, int total_rows, int deterministic, void* stream) {
  SERL_CHECK_ARG(mean && log_std_raw && act && B >= 1 && A >= 1, "tanh_gaussian_fwd: bad args");
  SERL_CHECK_ARG(row_offset >= 0 && row_offset + B <= total_rows, "tanh_gaussian_fwd: shard outside the global batch");
  TgArgs t{mean, log_std_raw, ld_ms, act, ld_act, xpre, eps, logp, B, A, k0, k1, row_offset, total_rows, deterministic};
  hipLaunchKernelGGL(tanh_gaussian_fwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, t);
  SERL_CHECK_LAUNCH("tanh_gaussian_fwd");
  return SERL_OK;
}

// Actor-loss backward through the head: L = mean_b(alpha*logp_b - q_b), da = dL/da from the critic path.
//   dL/dmean = (alpha/Bg)*2a + da*(1-a^2);  dL/dlogstd = (alpha/Bg)*(-1 + 2a*sigma*eps) + da*(1-a^2)*sigma*eps,
// zeroed where the raw log-std was clipped.  Output: dms (B, 2A) = [dmean | dlogstd] for the fused head GEMM.
__global__ void tanh_gaussian_bwd_kernel(const float* act, int ld_act, const float* eps, const float* lstd, int ld_ms,
                                         const float* da, int ld_da, const float* log_temp, float inv_batch,
                                         float* dmean, float* dlstd, int ld_out, int B, int A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * A) return;
  const int b = i / A, j = i - b * A;
  const float alpha = expf(log_temp[0]);
  const float a = act[(long long)b * ld_act + j];
  const float raw = lstd[(long long)b * ld_ms + j];
  const float ls = fminf(fmaxf(raw, -20.f), 2.f);
  const float se = expf(ls) * eps[i];
  const float dla = da[(long long)b * ld_da + j] * (1.f - a * a);
  const float c = alpha * inv_batch;
  dmean[(long long)b * ld_out + j] = c * 2.f * a + dla;
  const float dl = c * (-1.f + 2.f * a * se) + dla * se;
  dlstd[(long long)b * ld_out + j] = (raw < -20.f || raw > 2.f) ? 0.f : dl;
}
extern "C" int serl_tanh_gaussian_bwd(const float* act, int ld_act, const float* eps, const float* log_std_raw,
                                      int ld_ms, const float* da, int ld_da, const float* log_temp, float inv_batch,
                                      float* dmean, float* dlstd, int ld_out, int B, int A, void* stream) {
  SERL_CHECK_ARG(act && eps && log_std_raw && da && log_temp && dmean && dlstd, "tanh_gaussian_bwd: null pointer");
  hipLaunchKernelGGL(tanh_gaussian_bwd_kernel, dim3(cdiv(B * A, 256)), dim3(256), 0, (hipStream_t)stream, act, ld_act,
                     eps, log_std_raw, ld_ms, da, ld_da, log_temp, inv_batch, dmean, dlstd, ld_out, B, A);
  SERL_CHECK_LAUNCH("tanh_gaussian_bwd");
  return SERL_OK;
}

// ------------------------------------------------------------------ block reduction helper (<= 1024 threads)
__device__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nw; ++i) s += sh[i];
  return s;
}

// ------------------------------------------------------------------ behaviour-cloning losses
// nll = -mean_b log_prob(a_b) of the tanh-Gaussian policy for GIVEN actions (serl/agents/bc/pixel_bc_learner.py:114-134,
// pixel_hybrid_bc_learner.py:145-165): a clipped to [-1 + 1e-5, 1 - 1e-5], x = atanh(a) (TFP Tanh bijector inverse),
//   log_prob = sum_j [-z^2/2 - ls - log(2 pi)/2] - sum_j 2 (log 2 - x - softplus(-2x)),  z = (x - mu) exp(-ls).
// dL/dmu = -(z exp(-ls)) / Bg, dL/dls = -(z^2 - 1) / Bg (0 where the raw log-std was clipped).  info[0] = sum_b -log_prob.
__global__ __launch_bounds__(256) void tanh_gaussian_nll_kernel(const float* mean, const float* lstd, int ld_ms,
                                                                const float* act, int ld_act, int B, int A,
                                                                float inv_batch, float* dmean, float* dlstd, int ld_out,
                                                                float* info) {
  __shared__ float sh[4];
  float nll = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float lp = 0.f;
    for (int j = 0; j < A; ++j) {
      const float a = fminf(fmaxf(act[(long long)b * ld_act + j], -1.f + 1e-5f), 1.f - 1e-5f);
      const float x = atanhf(a);
      const float mu = mean[(long long)b * ld_ms + j];
      const float raw = lstd[(long long)b * ld_ms + j];
      const float ls = fminf(fmaxf(raw, -20.f), 2.f);
      const float is = expf(-ls), z = (x - mu) * is;
      lp += (-0.5f * z * z - ls - 0.918938533204672742f) - 2.f * (0.693147180559945309f - x - softplusf(-2.f * x));
      dmean[(long long)b * ld_out + j] = -z * is * inv_batch;
      dlstd[(long long)b * ld_out + j] = (raw < -20.f || raw > 2.f) ? 0.f : -(z * z - 1.f) * inv_batch;
    }
    nll -= lp;
  }
  nll = block_sum(nll, sh);
  if (threadIdx.x == 0) info[0] = nll;
}
extern "C" int serl_tanh_gaussian_nll(const float* mean, const float* log_std_raw, int ld_ms, const float* actions,
                                      int ld_act, int B, int A, float inv_batch, float* dmean, float* dlstd, int ld_out,
                                      float* info, void* stream) {
  SERL_CHECK_ARG(mean && log_std_raw && actions && dmean && dlstd && info && B >= 1 && A >= 1, "tanh_gaussian_nll: bad args");
  hipLaunchKernelGGL(tanh_gaussian_nll_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, mean, log_std_raw, ld_ms, actions,
                     ld_act, B, A, inv_batch, dmean, dlstd, ld_out, info);
  SERL_CHECK_LAUNCH("tanh_gaussian_nll");
  return SERL_OK;
}

// log pi(a) of GIVEN actions under the tanh-Gaussian, one thread per row: TanhTransformedDistribution.log_prob
// (serl/distributions/tanh_transformed.py:17-30 over tanh_normal.py:41-48): x = atanh(clip(a, +-nextafter(1, 0))) -- the
// Tanh bijector's inverse in float32 -- then the diagonal-normal log-density minus the Tanh log-det-Jacobian.
__global__ void tanh_gaussian_log_prob_kernel(const float* mean, const float* lstd, int ld_ms, const float* act, int ld_act,
                                              float* logp, int B, int A) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float lp = 0.f;
  for (int j = 0; j < A; ++j) {
    const float a = fminf(fmaxf(act[(long long)b * ld_act + j], -0.99999994f), 0.99999994f);
    const float x = atanhf(a);
    const float mu = mean[(long long)b * ld_ms + j];
    const float ls = fminf(fmaxf(lstd[(long long)b * ld_ms + j], -20.f), 2.f);
    const float z = (x - mu) * expf(-ls);
    lp += (-0.5f * z * z - ls - 0.918938533204672742f) - 2.f * (0.693147180559945309f - x - softplusf(-2.f * x));
  }
  logp[b] = lp;
}
extern "C" int serl_tanh_gaussian_log_prob(const float* mean, const float* log_std_raw, int ld_ms, const float* actions,
                                           int ld_act, float* logp, int B, int A, void* stream) {
  SERL_CHECK_ARG(mean && log_std_raw && actions && logp && B >= 1 && A >= 1 && ld_ms >= A && ld_act >= A, "tanh_gaussian_log_prob: bad args");
  hipLaunchKernelGGL(tanh_gaussian_log_prob_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, mean, log_std_raw, ld_ms,
                     actions, ld_act, logp, B, A);
  SERL_CHECK_LAUNCH("tanh_gaussian_log_prob");
  return SERL_OK;
}

// optax.sigmoid_binary_cross_entropy(logits, labels).mean() with labels = (a > 0) of one action column
// (pixel_hybrid_bc_learner.py:167-176): l = -y log_sigmoid(x) - (1 - y) log_sigmoid(-x) = max(x, 0) - x y + log1p(exp(-|x|)),
// dl/dx = sigmoid(x) - y.  info[0] = sum_b l_b, dlogit = dl/dx / Bg.
__global__ __launch_bounds__(256) void bce_logits_kernel(const float* logits, const float* act, int ld_act, int B,
                                                         float inv_batch, float* dlogit, float* info) {
  __shared__ float sh[4];
  float l = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const float x = logits[b], y = act[(long long)b * ld_act] > 0.f ? 1.f : 0.f;
    l += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
    dlogit[b] = (1.f / (1.f + expf(-x)) - y) * inv_batch;
  }
  l = block_sum(l, sh);
  if (threadIdx.x == 0) info[0] = l;
}
extern "C" int serl_bce_logits(const float* logits, const float* action_column, int ld_act, int B, float inv_batch,
                               float* dlogit, float* info, void* stream) {
  SERL_CHECK_ARG(logits && action_column && dlogit && info && B >= 1, "bce_logits: bad args");
  hipLaunchKernelGGL(bce_logits_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, action_column, ld_act, B, inv_batch,
                     dlogit, info);
  SERL_CHECK_LAUNCH("bce_logits");
  return SERL_OK;
}

// ------------------------------------------------------------------ TD target + critic loss gradient
// target = r + discount*mask*min_e next_q[e] (- discount*mask*alpha*next_logp if backup_entropy)
// loss = mean_{e,b} (q - target)^2 ; dq = 2 (q - target) * inv_count  (sac_learner.py:185-209)
// info_sums[0] += sum (q-t)^2, info_sums[1] += sum q  (divided on the host by the global E*B).
__global__ __launch_bounds__(256) void td_loss_kernel(const float* next_q, int Emin, const float* rew, const float* mask,
                                                      const float* next_logp, const float* log_temp, float discount,
                                                      const float* q, int E, int B, float inv_count, float* target,
                                                      float* dq, float* info) {
  __shared__ float sh[4];
  float l = 0.f, qs = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float mn = next_q[b];
    for (int e = 1; e < Emin; ++e) mn = fminf(mn, next_q[e * B + b]);
    float t = rew[b] + discount * mask[b] * mn;
    if (next_logp) t -= discount * mask[b] * expf(log_temp[0]) * next_logp[b];
    target[b] = t;
    for (int e = 0; e < E; ++e) {
      const float qv = q[e * B + b], d = qv - t;
      dq[e * B + b] = 2.f * d * inv_count;
      l += d * d;
      qs += qv;
    }
  }
  l = block_sum(l, sh);
  qs = block_sum(qs, sh);
  if (threadIdx.x == 0) { info[0] = l; info[1] = qs; }
}
extern "C" int serl_td_loss(const float* next_q, int Emin, const float* rew, const float* mask, const float* next_logp,
                            const float* log_temp, float discount, const float* q, int E, int B, float inv_count,
                            float* target, float* dq, float* info, void* stream) {
  SERL_CHECK_ARG(next_q && rew && mask && q && target && dq && info && Emin >= 1 && E >= 1 && B >= 1, "td_loss: bad args");
  SERL_CHECK_ARG(!next_logp || log_temp, "td_loss: entropy backup needs log_temp");
  hipLaunchKernelGGL(td_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, next_q, Emin, rew, mask, next_logp,
                     log_temp, discount, q, E, B, inv_count, target, dq, info);
  SERL_CHECK_LAUNCH("td_loss");
  return SERL_OK;
}

// actor loss sums: info[0] = sum_b (alpha*logp_b - mean_e q[e,b]), info[1] = sum_b logp_b  (sac_learner.py:135-139)
__global__ __launch_bounds__(256) void actor_loss_kernel(const float* q, int E, int B, const float* logp,
                                                         const float* log_temp, float* info) {
  __shared__ float sh[4];
  const float alpha = expf(log_temp[0]);
  float l = 0.f, lp = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float qm = 0.f;
    for (int e = 0; e < E; ++e) qm += q[e * B + b];
    qm /= (float)E;
    l += logp[b] * alpha - qm;
    lp += logp[b];
  }
  l = block_sum(l, sh);
  lp = block_sum(lp, sh);
  if (threadIdx.x == 0) { info[0] = l; info[1] = lp; }
}
extern "C" int serl_actor_loss(const float* q, int E, int B, const float* logp, const float* log_temp, float* info,
                               void* stream) {
  SERL_CHECK_ARG(q && logp && log_temp && info && E >= 1 && B >= 1, "actor_loss: bad args");
  hipLaunchKernelGGL(actor_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, q, E, B, logp, log_temp, info);
  SERL_CHECK_LAUNCH("actor_loss");
  return SERL_OK;
}

// ------------------------------------------------------------------ fused Adam (+ Polyak target) over a flat buffer
// optax.adam: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr * (m/bc1) / (sqrt(v/bc2) + eps)
// optax.incremental_update: target = tau * p_new + (1 - tau) * target   (sac_learner.py:212-217)
__global__ __launch_bounds__(256) void adam_polyak_kernel(float* __restrict__ p, float* __restrict__ m,
                                                          float* __restrict__ v, const float* __restrict__ g,
                                                          float* __restrict__ tgt, long long n, float lr, float b1,
                                                          float b2, float eps, float bc1, float bc2, float tau,
                                                          float gscale) {
  const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  if (i4 + 4 <= n) {
    float4 P = *(float4*)(p + i4), M = *(float4*)(m + i4), V = *(float4*)(v + i4);
    const float4 G = *(const float4*)(g + i4);
    float4 T;
    if (tgt) T = *(float4*)(tgt + i4);
    float* pp = (float*)&P; float* mm = (float*)&M; float* vv = (float*)&V; const float* gg = (const float*)&G;
    float* tt = (float*)&T;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = gg[k] * gscale;
      mm[k] = b1 * mm[k] + (1.f - b1) * gk;
      vv[k] = b2 * vv[k] + (1.f - b2) * gk * gk;
      pp[k] = pp[k] - lr * ((mm[k] / bc1) / (sqrtf(vv[k] / bc2) + eps));
      if (tgt) tt[k] = tau * pp[k] + (1.f - tau) * tt[k];
    }
    *(float4*)(p + i4) = P; *(float4*)(m + i4) = M; *(float4*)(v + i4) = V;
    if (tgt) *(float4*)(tgt + i4) = T;
  } else {
    for (long long i = i4; i < n; ++i) {
      const float gk = g[i] * gscale;
      m[i] = b1 * m[i] + (1.f - b1) * gk;
      v[i] = b2 * v[i] + (1.f - b2) * gk * gk;
      p[i] = p[i] - lr * ((m[i] / bc1) / (sqrtf(v[i] / bc2) + eps));
      if (tgt) tgt[i] = tau * p[i] + (1.f - tau) * tgt[i];
    }
  }
}
extern "C" int serl_adam_polyak(float* p, float* m, float* v, const float* g, float* target, long long n, float lr,
                                float b1, float b2, float eps, float bc1, float bc2, float tau, float grad_scale,
                                void* stream) {
  SERL_CHECK_ARG(p && m && v && g && n >= 1, "adam_polyak: bad args");
  SERL_CHECK_ARG((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g | (uintptr_t)target) & 15) == 0,
                 "adam_polyak: buffers must be 16-byte aligned");
  const long long nthreads = cdivll(n, 4);
  hipLaunchKernelGGL(adam_polyak_kernel, dim3((unsigned)cdivll(nthreads, 256)), dim3(256), 0, (hipStream_t)stream, p, m,
                     v, g, target, n, lr, b1, b2, eps, bc1, bc2, tau, grad_scale);
  SERL_CHECK_LAUNCH("adam_polyak");
  return SERL_OK;
}

// Temperature: loss = alpha*(entropy - target_entropy); d/dlog_temp = same value; scalar Adam on device
// (sac_learner.py:146-158).  entropy = -sum_logp * inv_batch is read from device memory (no host sync).
// state = [log_temp, m, v]; info_out = [temperature, temperature_loss, entropy].
__global__ void temperature_kernel(float* state, const float* sum_logp, float inv_batch, float target_entropy, float lr,
                                   float b1, float b2, float eps, float bc1, float bc2, float* info_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float entropy = -sum_logp[0] * inv_batch;
  const float temp = expf(state[0]);
  const float g = temp * (entropy - target_entropy);
  info_out[0] = temp; info_out[1] = g; info_out[2] = entropy;
  const float m = b1 * state[1] + (1.f - b1) * g;
  const float v = b2 * state[2] + (1.f - b2) * g * g;
  state[0] = state[0] - lr * ((m / bc1) / (sqrtf(v / bc2) + eps));
  state[1] = m; state[2] = v;
}
extern "C" int serl_temperature_update(float* state, const float* sum_logp, float inv_batch, float target_entropy,
                                       float lr, float b1, float b2, float eps, float bc1, float bc2, float* info_out,
                                       void* stream) {
  SERL_CHECK_ARG(state && sum_logp && info_out, "temperature_update: null pointer");
  hipLaunchKernelGGL(temperature_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, sum_logp, inv_batch,
                     target_entropy, lr, b1, b2, eps, bc1, bc2, info_out);
  SERL_CHECK_LAUNCH("temperature_update");
  return SERL_OK;
}

// ------------------------------------------------------------------ nn.Dropout (training) on a (rows, cols) block
// x[r][c] = keep ? x[r][c] / keep_prob : 0 with keep = jax.random.bernoulli(key, keep_prob, (total_rows, cols))
// [row_offset + r][c] (uniform(key) < keep_prob; threefry counter = global row * cols + c, so rank shards and the
// backward pass -- the same call on the gradient -- reproduce the mask without storing it).
// serl/networks/encoders/two_mobilenet_encoder.py:34, serl/networks/encoded_encoder.py:34.
__global__ __launch_bounds__(256) void dropout_kernel(float* x, int ld, int rows, int cols, uint32_t k0, uint32_t k1,
                                                      float keep_prob, int row_offset, uint32_t total) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
  // row_offset < 0: every row takes row 0 of a (1, cols) draw (the mask vmap(grad(...)) with a closed-over rng applies)
  const uint32_t gr = row_offset < 0 ? 0u : (uint32_t)(row_offset + r);
  const uint32_t bits = threefry_bits_at(k0, k1, gr * (uint32_t)cols + (uint32_t)c, total);
  const float u = __uint_as_float((bits >> 9) | 0x3F800000u) - 1.0f;
  float* p = x + (long long)r * ld + c;
  *p = u < keep_prob ? *p / keep_prob : 0.f;
}
extern "C" int serl_dropout(float* x, int ld, int rows, int cols, unsigned k0, unsigned k1, float keep_prob, int row_offset,
                            int total_rows, void* stream) {
  SERL_CHECK_ARG(x && rows >= 1 && cols >= 1 && ld >= cols && keep_prob > 0.f && keep_prob <= 1.f, "dropout: bad args");
  SERL_CHECK_ARG((row_offset == -1 && total_rows == 1) || (row_offset >= 0 && row_offset + rows <= total_rows),
                 "dropout: rows outside the global draw");
  SERL_CHECK_ARG((long long)total_rows * cols < (1ll << 32), "dropout: more than 2^32 elements");
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)cdivll((long long)rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     ld, rows, cols, k0, k1, keep_prob, row_offset, (uint32_t)((long long)total_rows * cols));
  SERL_CHECK_LAUNCH("dropout");
  return SERL_OK;
}

// ------------------------------------------------------------------ VICE reward (vice_learner.py:274,301)
// mean[i] = mean over the cameras of sigmoid(logits[c][i]); rewards[i] = log(mean / (1 - mean)); one thread per sample.
struct ViceRewArgs { const float* logits[4]; float* sig[4]; int ncam; };
__global__ void vice_reward_kernel(const ViceRewArgs a, float* mean_out, float* rewards, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float m = 0.f;
  for (int c = 0; c < a.ncam; ++c) {
    const float s = 1.f / (1.f + expf(-a.logits[c][i]));
    if (a.sig[c]) a.sig[c][i] = s;
    m += s;
  }
  m /= (float)a.ncam;
  if (mean_out) mean_out[i] = m;
  if (rewards) rewards[i] = logf(m / (1.f - m));
}
extern "C" int serl_vice_reward(int ncam, const float* const* logits, float* const* sigmoids, float* mean_out,
                                float* rewards, int n, void* stream) {
  SERL_CHECK_ARG(ncam >= 1 && ncam <= 4 && logits && n >= 1, "vice_reward: bad args");
  ViceRewArgs a;
  a.ncam = ncam;
  for (int c = 0; c < ncam; ++c) {
    SERL_CHECK_ARG(logits[c], "vice_reward: null logits");
    a.logits[c] = logits[c];
    a.sig[c] = sigmoids ? sigmoids[c] : nullptr;
  }
  hipLaunchKernelGGL(vice_reward_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a, mean_out, rewards, n);
  SERL_CHECK_LAUNCH("vice_reward");
  return SERL_OK;
}
