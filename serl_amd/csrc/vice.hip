// Kernels of VICELearner.update_classifier (serl/agents/vice/vice_learner.py:316-474): mixup of the encoded batch, the
// mixup binary cross-entropy, and the gradient penalty 10 * mean((||d f / d x|| - 1)^2), whose parameter gradient the
// reference gets from jax.grad over vmap(jax.grad(...)) (:453-467).  Here the double backward is hand-derived as
// "reverse mode through the tangent pass": with G_i = d y_i / d x_i held constant,
//     d(0.5 ||G_i||^2) / d theta = d/d theta <d y_i / d x_i, G_i> = d/d theta (JVP of y_i along G_i),
// so the penalty's gradient is the ordinary reverse-mode gradient of  Phi = sum_i alpha_i * ydot_i  where ydot is the
// forward tangent of the classifier along G and alpha_i = 10 * 2 (n_i - 1) / (M n_i), n_i = sqrt(||G_i||^2 + D * 1e-6).
// The tangent of LayerNorm + activation and its adjoint (w.r.t. both the tangent input and the primal input) are the
// two row kernels below; everything else is GEMMs, column sums and the SLE kernels.  One wave (64 lanes) per row.
#include "../../include/serl_hip.h"
#include "common.h"

namespace {
constexpr int MAXW = 16;   // row width <= 1024
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_LEAKY = 3 };

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float act_prime(float y, int act) {
  if (act == ACT_TANH) return 1.f - y * y;
  if (act == ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == ACT_LEAKY) return y > 0.f ? 1.f : 0.01f;
  return 1.f;
}

// out[r][:] = wa[r] * x[ia[r]][:] + wb[r] * x[ib[r]][:]   (mixup :361, gradient-penalty interpolation :428)
__global__ __launch_bounds__(256) void row_combine_kernel(const float* x, const int* ia, const int* ib, const float* wa,
                                                          const float* wb, float* out, int rows, long long cols4, int src_rows) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols4) return;
  const int r = (int)(i / cols4);
  const long long c = i - (long long)r * cols4;
  // source rows are clamped to [0, src_rows): a bad index table gives a wrong row, never an out-of-bounds read
  const int ra = min(max(ia[r], 0), src_rows - 1), rb = min(max(ib[r], 0), src_rows - 1);
  const float4 a = ((const float4*)x)[(long long)ra * cols4 + c], b = ((const float4*)x)[(long long)rb * cols4 + c];
  const float u = wa[r], v = wb[r];
  ((float4*)out)[i] = make_float4(u * a.x + v * b.x, u * a.y + v * b.y, u * a.z + v * b.z, u * a.w + v * b.w);
}

// mixup BCE (:443-449): loss = sum_i [lam * bce(y_i, ya_i) + (1 - lam) * bce(y_i, yb_i)] / n, bce(y, l) = softplus(y) - l y;
// dy_i = (sigmoid(y_i) - (lam ya_i + (1 - lam) yb_i)) / n.  One block.
__global__ __launch_bounds__(256) void bce_mixup_kernel(const float* y, const float* ya, const float* yb, float lam, int n,
                                                        float* dy, float* loss) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float v = y[i], l = lam * ya[i] + (1.f - lam) * yb[i];
    const float sp = fmaxf(v, 0.f) + log1pf(expf(-fabsf(v)));
    s += sp - l * v;
    dy[i] = (1.f / (1.f + expf(-v)) - l) / (float)n;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

// gradient-penalty row statistics (:461-465): n_i = sqrt(sum_j (G_ij^2 + 1e-6)); alpha_i = scale * 2 (n_i - 1) / (M n_i);
// stats[0] += n_i / M (grad_norm), stats[1] += (n_i - 1)^2 / M (penalty).  One block per row; stats zeroed by the caller.
__global__ __launch_bounds__(256) void gp_norm_kernel(const float* G, long long D, int M, float scale, float* alpha, float* norms) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  float s = 0.f;
  for (long long j = threadIdx.x; j < D; j += 256) { const float g = G[(long long)r * D + j]; s += g * g + 1e-6f; }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float n = sqrtf(red[0] + red[1] + red[2] + red[3]);
    norms[r] = n;
    alpha[r] = scale * 2.f * (n - 1.f) / ((float)M * n);
  }
}

// ---- tangent of [LayerNorm] + activation.  Primal: h = xhat (normalised input), rstd = 1 / sigma, y = act(gamma h + beta).
// Tangent input a (R, N) -> w = (a - mean(a) - h mean(h a)) rstd (w = a without LN); ndot = gamma w; out = act'(y) ndot.
struct TanFwd { const float* a; const float* xhat; const float* rstd; const float* gamma; const float* y; float* out; float* w;
                int R, N, use_ln, act; };
__global__ __launch_bounds__(256) void ln_tangent_fwd_kernel(const TanFwd t) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= t.R) return;
  const long long o = (long long)row * t.N;
  float a[MAXW], h[MAXW];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int n = lane + 64 * i;
    a[i] = 0.f; h[i] = 0.f;
    if (n < t.N) {
      a[i] = t.a[o + n];
      if (t.use_ln) { h[i] = t.xhat[o + n]; s1 += a[i]; s2 += h[i] * a[i]; }
    }
  }
  float c1 = 0.f, c2 = 0.f, rs = 1.f;
  if (t.use_ln) { c1 = wave_sum(s1) / (float)t.N; c2 = wave_sum(s2) / (float)t.N; rs = t.rstd[row]; }
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int n = lane + 64 * i;
    if (n < t.N) {
      const float w = t.use_ln ? (a[i] - c1 - h[i] * c2) * rs : a[i];
      const float nd = t.use_ln ? t.gamma[n] * w : w;
      t.w[o + n] = w;
      t.out[o + n] = act_prime(t.y[o + n], t.act) * nd;
    }
  }
}

// ---- adjoint of the above, for Phi = sum alpha ydot.  Inputs: outbar (adjoint of `out`), tbar_ext (adjoint arriving at
// the PRIMAL activated output y from later layers; may be NULL).  Outputs: abar (adjoint of the tangent input a), ubar
// (adjoint of the primal pre-LayerNorm input), ggamma (per-row terms of the scale gradient: ndotbar w + nbar h), nbar
// (per-row terms of the shift gradient).  Derivation in DESIGN.md ("gradient penalty").
struct TanBwd { const float* outbar; const float* tbar_ext; const float* y; const float* w; const float* a; const float* xhat;
                const float* rstd; const float* gamma; float* abar; float* ubar; float* ggamma; float* nbar;
                int R, N, use_ln, act; };
__global__ __launch_bounds__(256) void ln_tangent_bwd_kernel(const TanBwd t) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= t.R) return;
  const long long o = (long long)row * t.N;
  float wbar[MAXW], nb[MAXW], h[MAXW], a[MAXW], w[MAXW];
  float m1 = 0.f, m2 = 0.f, c2 = 0.f, sw = 0.f;
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int n = lane + 64 * i;
    wbar[i] = nb[i] = h[i] = a[i] = w[i] = 0.f;
    if (n < t.N) {
      const float y = t.y[o + n], ob = t.outbar[o + n], ap = act_prime(y, t.act);
      w[i] = t.w[o + n];
      const float g = t.use_ln ? t.gamma[n] : 1.f;
      const float nd = g * w[i];
      const float ndbar = ap * ob;
      const float extra = t.act == ACT_TANH ? -2.f * y * nd * ob : 0.f;      // d/dy of act'(y) = 1 - y^2
      nb[i] = ap * ((t.tbar_ext ? t.tbar_ext[o + n] : 0.f) + extra);
      wbar[i] = g * ndbar;
      a[i] = t.a[o + n];
      if (t.use_ln) {
        h[i] = t.xhat[o + n];
        t.ggamma[o + n] = ndbar * w[i] + nb[i] * h[i];
        t.nbar[o + n] = nb[i];
        m1 += wbar[i]; m2 += wbar[i] * h[i]; c2 += h[i] * a[i]; sw += wbar[i] * w[i];
      } else {
        t.abar[o + n] = ndbar;
        t.ubar[o + n] = nb[i];
      }
    }
  }
  if (!t.use_ln) return;
  const float invN = 1.f / (float)t.N, rs = t.rstd[row];
  m1 = wave_sum(m1) * invN; m2 = wave_sum(m2) * invN; c2 = wave_sum(c2) * invN; sw = wave_sum(sw);
  const float sigbar = -sw * rs;                                  // adjoint of sigma from the explicit 1 / sigma of w
  float hb[MAXW];
  float m3 = 0.f, m4 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int n = lane + 64 * i;
    hb[i] = 0.f;
    if (n < t.N) {
      t.abar[o + n] = (wbar[i] - m1 - h[i] * m2) * rs;
      hb[i] = -(wbar[i] * c2 + a[i] * m2) * rs + t.gamma[n] * nb[i];   // explicit h in w, plus the primal path n = gamma h + beta
      m3 += hb[i]; m4 += hb[i] * h[i];
    }
  }
  m3 = wave_sum(m3) * invN; m4 = wave_sum(m4) * invN;
#pragma unroll
  for (int i = 0; i < MAXW; ++i) {
    const int n = lane + 64 * i;
    if (n < t.N) t.ubar[o + n] = (hb[i] - m3 - h[i] * m4) * rs + sigbar * h[i] * invN;
  }
}

__global__ void axpy_kernel(float* y, const float* x, float a, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += a * x[i];
}
}  // namespace

extern "C" int serl_row_combine(const float* x, int src_rows, const int* ia, const int* ib, const float* wa, const float* wb,
                                float* out, int rows, long long cols, void* stream) {
  SERL_CHECK_ARG(x && ia && ib && wa && wb && out && rows >= 1 && src_rows >= 1 && cols >= 4 && cols % 4 == 0, "row_combine: bad args");
  SERL_CHECK_ARG((((uintptr_t)x | (uintptr_t)out) & 15) == 0, "row_combine: buffers must be 16-byte aligned");
  const long long total = (long long)rows * (cols / 4);
  hipLaunchKernelGGL(row_combine_kernel, dim3((unsigned)cdivll(total, 256)), dim3(256), 0, (hipStream_t)stream, x, ia, ib, wa,
                     wb, out, rows, cols / 4, src_rows);
  SERL_CHECK_LAUNCH("row_combine");
  return SERL_OK;
}
extern "C" int serl_bce_mixup(const float* y, const float* ya, const float* yb, float lam, int n, float* dy, float* loss,
                              void* stream) {
  SERL_CHECK_ARG(y && ya && yb && dy && loss && n >= 1, "bce_mixup: bad args");
  hipLaunchKernelGGL(bce_mixup_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, y, ya, yb, lam, n, dy, loss);
  SERL_CHECK_LAUNCH("bce_mixup");
  return SERL_OK;
}
extern "C" int serl_gp_norm(const float* G, int rows, long long D, float scale, float* alpha, float* norms, void* stream) {
  SERL_CHECK_ARG(G && alpha && norms && rows >= 1 && D >= 1, "gp_norm: bad args");
  hipLaunchKernelGGL(gp_norm_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, G, D, rows, scale, alpha, norms);
  SERL_CHECK_LAUNCH("gp_norm");
  return SERL_OK;
}
extern "C" int serl_ln_tangent_fwd(const float* a, const float* xhat, const float* rstd, const float* gamma, const float* y,
                                   float* out, float* w, int R, int N, int use_ln, int act, void* stream) {
  SERL_CHECK_ARG(a && y && out && w && R >= 1 && N >= 1 && N <= 64 * MAXW, "ln_tangent_fwd: bad args");
  SERL_CHECK_ARG(!use_ln || (xhat && rstd && gamma), "ln_tangent_fwd: LayerNorm needs xhat, rstd, scale");
  TanFwd t{a, xhat, rstd, gamma, y, out, w, R, N, use_ln, act};
  hipLaunchKernelGGL(ln_tangent_fwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, t);
  SERL_CHECK_LAUNCH("ln_tangent_fwd");
  return SERL_OK;
}
extern "C" int serl_ln_tangent_bwd(const float* outbar, const float* tbar_ext, const float* y, const float* w, const float* a,
                                   const float* xhat, const float* rstd, const float* gamma, float* abar, float* ubar,
                                   float* ggamma, float* nbar, int R, int N, int use_ln, int act, void* stream) {
  SERL_CHECK_ARG(outbar && y && w && a && abar && ubar && R >= 1 && N >= 1 && N <= 64 * MAXW, "ln_tangent_bwd: bad args");
  SERL_CHECK_ARG(!use_ln || (xhat && rstd && gamma && ggamma && nbar), "ln_tangent_bwd: LayerNorm needs xhat, rstd, scale, ggamma, nbar");
  TanBwd t{outbar, tbar_ext, y, w, a, xhat, rstd, gamma, abar, ubar, ggamma, nbar, R, N, use_ln, act};
  hipLaunchKernelGGL(ln_tangent_bwd_kernel, dim3(cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, t);
  SERL_CHECK_LAUNCH("ln_tangent_bwd");
  return SERL_OK;
}
extern "C" int serl_axpy(float* y, const float* x, float a, long long n, void* stream) {
  SERL_CHECK_ARG(y && x && n >= 1, "axpy: bad args");
  hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)cdivll(n, 256)), dim3(256), 0, (hipStream_t)stream, y, x, a, n);
  SERL_CHECK_LAUNCH("axpy");
  return SERL_OK;
}
