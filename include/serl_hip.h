/* libserl_hip.so -- C ABI of the MI355X (gfx950) kernels behind serl_amd.
 *
 * The reference (serl-robot/serl) has no FFI on this path: its learner update is Python that XLA
 * compiles.  The drop-in boundary therefore is the Python agent/buffer API; this
 * header is the thin kernel layer underneath it that `serl_amd/_lib.py` binds with ctypes.  Every entry
 * point cites the reference code whose arithmetic it replaces (paths relative to the reference root).
 *
 * Conventions: plain pointers and sizes only; all device pointers are caller-allocated; `stream` is a
 * hipStream_t; tensors are row-major float32 unless stated; images are NHWC, conv kernels HWIO (flax).
 * Return value: 0 ok, 1 invalid argument (-> ValueError), 2 HIP error (-> RuntimeError); the message is in
 * serl_last_error() (thread-local).  The kernel entry points neither allocate device memory nor synchronise; the
 * exceptions are the step driver (serl_learner_*: the handle owns HIP streams, events and a lazily allocated activation
 * workspace, see there) and the first call of an entry point on a device, which may query device attributes.
 */
#ifndef SERL_HIP_H
#define SERL_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

const char* serl_last_error(void);
int serl_version(void);
int serl_device_arch(int device, char* out, int out_len);

/* Process-wide precision of every conv / GEMM entry point below; the counterpart of the reference's
 * jax.config "jax_default_matmul_precision" (the reference never sets it: fp32 on CPU, TF32 on NVIDIA GPUs).
 *   0 "highest": v_mfma_f32_32x32x2_f32, exact fp32 products, fp32 accumulate.
 *   1 "bf16_3x" (default): fp32 operands split in-kernel into bf16 hi + lo, three v_mfma_f32_32x32x16_bf16
 *     per product block (lo*hi + hi*lo + hi*hi), fp32 accumulate: <= 2^-16 relative error per product. */
int serl_set_matmul_precision(int mode);
int serl_get_matmul_precision(void);

/* ---- convolution encoder: flax nn.Conv(VALID, bias) + ReLU, serl/networks/encoders/two_d4pg_encoder.py:30-38.
 * Grouped: group g has its own x/w/bias/y pointers (cameras x forward passes), <= 8 groups.
 * in_u8: x is uint8 and is scaled by 1/255 (correctly rounded) on load, serl/networks/pixel_multiplexer.py:27. */
int serl_conv2d_fwd(int ngroups, const void* const* x, const float* const* w, const float* const* bias,
                    float* const* y, int in_u8, int B, int IH, int IW, int C, int KH, int KW, int stride, int F,
                    int relu, void* stream);
/* dX = conv-transpose(dZ, W) masked by (xact > 0): the backward XLA derives for the same conv + ReLU stack. */
int serl_conv2d_dgrad(int ngroups, const float* const* dz, const float* const* w, const float* const* xact,
                      float* const* dx, int B, int IH, int IW, int C, int KH, int KW, int stride, int F, void* stream);
/* dW, dbias of the same conv (split-K slabs in `workspace`, summed in a fixed order: reproducible). */
long long serl_conv2d_wgrad_workspace_bytes(int ngroups, int B, int IH, int IW, int C, int KH, int KW, int stride,
                                            int F);
int serl_conv2d_wgrad(int ngroups, const void* const* x, const float* const* dz, float* const* dw,
                      float* const* dbias, float* workspace, long long workspace_bytes, int in_u8, int B, int IH,
                      int IW, int C, int KH, int KW, int stride, int F, void* stream);

/* ---- the same convolutions on S-format tensors (precision "bf16_3x" without any in-kernel conversion).
 * S format of a [rows][cols] fp32 tensor, cols % 8 == 0: unsigned short[rows][2][cols]; per row the cols bf16 "hi"
 * values (bf16_rn(x)) then the cols bf16 "lo" values (bf16_rn(x - hi)); 4 bytes per element like fp32, 16-byte
 * aligned.  Activations (NHWC: rows = pixels, cols = channels) and their gradients stay in this format between
 * layers; the first layer reads uint8 pixels (KW*C <= 32).  serl_convs_prep_weights re-lays an HWIO fp32 kernel
 * out as wf (forward operand, serl_convs_weight_bytes bytes) and wd (data-gradient operand, 4*KH*KW*C*F bytes;
 * may be NULL).  For a 3x3 layer with C = 128, F = 256 (the encoder's last layer) or C = 64, F = 128 (the one before)
 * serl_convs_weight_bytes is twice the S-format size: behind the [F][2][K] image the same call writes the weights once
 * more in the MFMA-fragment order of the kernel that serl_convs_fwd runs for that shape (csrc/pconv4.h streams them from
 * L2, pconv_fwdw_kernel loads them into its registers as one span per wave).  No counterpart in the reference: XLA picks its own internal
 * layouts. */
long long serl_convs_weight_bytes(int in_u8, int KH, int KW, int C, int F);
int serl_split_f32(const float* in, void* out, long long rows, int cols, void* stream);
int serl_merge_split(const void* in, float* out, long long rows, int cols, void* stream);
int serl_convs_prep_weights(int ngroups, const float* const* w, void* const* wf, void* const* wd, int in_u8, int KH,
                            int KW, int C, int F, void* stream);
/* the same for njobs <= 32 (layer, group) pairs in one launch; argument tables are host arrays of length njobs */
int serl_convs_prep_weights_multi(int njobs, const float* const* w, void* const* wf, void* const* wd, const int* in_u8,
                                  const int* KH, const int* KW, const int* C, const int* F, void* stream);
int serl_convs_fwd(int ngroups, const void* const* x, const void* const* wf, const float* const* bias, void* const* y,
                   int in_u8, int B, int IH, int IW, int C, int KH, int KW, int stride, int F, int relu,
                   void* const* rmask, void* stream);
/* rmask (may be NULL, entries may be NULL): per group a byte mask [B*OH*OW][F/8] of y, bit j of byte [pixel][n/8] =
 * "y[pixel][n+j] != 0" -- the ReLU mask for the data gradient of the next layer, 32x smaller than y's hi plane.
 * serl_convs_dgrad masks dx with xact: the layer's post-ReLU input in S format (xact_bits = 0), or that byte mask
 * (xact_bits = 1); xact NULL or a NULL entry: no mask. */
/* 3x3 stride-2 layers whose rows fit are run by the "patch" kernels (csrc/pconv.h: a band of input rows staged once in
 * LDS, all taps read from there; C = 64 <-> F = 128, forward and data gradient, by four waves that hold the whole
 * weight matrix in registers;
 * bit-identical output).  enable = 0 keeps every layer on the im2col kernels -- for A/B measurements and the bit-identity
 * test of the paths; 2 = without the 4-wave kernels; 3 = plus the LDS-DMA ring GEMM (csrc/sgemm_ring.h) for the deep
 * layers; 4 = serl_convs_fwd12 on its one-set kernel (default: two wave sets half a period apart); 5 / 6 / 7 = as 1 but
 * serl_convs_dgrad_wgrad1_supported / serl_convs_fwd12_keep_supported answer 0 / the C = 64 -> F = 128 weight gradient stays
 * on the im2col kernel; 8 = as 1 with the last layer's forward (15 x 15 x 128 -> 7 x 7 x 256, even B) on the im2col
 * kernel instead of pconv4_fwd_kernel, whose K loop runs in another order: equal to one step of the S format's 16-bit
 * mantissa, not bit for bit (A/B of the round-3 kernels).  Process-wide; no counterpart in the reference (XLA picks its
 * convolution algorithm itself, two_d4pg_encoder.py:30-38). */
int serl_convs_set_patch_kernels(int enable);
/* Persistent patch kernels launched after this call spread over at most `cus` compute units (0: all).  The step driver
 * brackets the actor's run-ahead pass with it (SERL_AMD_ACTOR_CUS).  Per calling thread (thread-local: a thread that
 * acts while another one learns is not affected); the weight gradient's grid -- its summation order -- ignores it.  No
 * counterpart in the reference. */
int serl_convs_set_cu_limit(int cus);
/* Layers 1 + 2 of a NO-GRADIENT encoder pass (target critic, actor: two_d4pg_encoder.py:30-38 under stop_gradient /
 * in the target network) in one launch: uint8 frames [B][IH1][IW1][C1] -> 3x3 stride-2 conv (-> F1 = 32) + bias + ReLU ->
 * 3x3 stride-2 conv (-> F2 = 64) + bias [+ ReLU] -> y (S format); wf1 / wf2 as serl_convs_prep_weights writes them
 * (in_u8 = 1 / 0).  Layer 1's output is never written to memory.  Bit-identical to the two serl_convs_fwd calls.
 * serl_convs_fwd12_supported returns 1 when the geometry fits the kernel (else serl_convs_fwd12 returns status 1). */
int serl_convs_fwd12_supported(int IH1, int IW1, int C1, int F1, int F2);
int serl_convs_fwd12(int ngroups, const void* const* x_u8, const void* const* wf1, const float* const* bias1,
                     const void* const* wf2, const float* const* bias2, void* const* y, int B, int IH1, int IW1, int C1,
                     int F1, int F2, int relu, void* const* rmask, void* stream);
/* serl_convs_fwd12 for the pass whose backward follows (the critic's / the actor's own pass): the groups with
 * y1[i] != NULL also get layer 1's output (S format [B][IH][IW][32], the bits serl_convs_fwd(in_u8 = 1) stores; rows
 * 0 .. 2 OH, i.e. every row layer 2 reads) and, with rmask1[i], its ReLU byte mask -- what the weight gradient of layer 2
 * and serl_convs_dgrad_wgrad1 need.  Layer 1's output is written once from the kernel's LDS patch and not read back by
 * layer 2 (0.26 GB per 2-camera pass at the headline size).  serl_convs_fwd12_keep_supported: geometry as
 * serl_convs_fwd12_supported, two-set kernel enabled. */
int serl_convs_fwd12_keep_supported(int IH1, int IW1, int C1, int F1, int F2);
int serl_convs_fwd12_keep(int ngroups, const void* const* x_u8, const void* const* wf1, const float* const* bias1,
                          const void* const* wf2, const float* const* bias2, void* const* y, int B, int IH1, int IW1,
                          int C1, int F1, int F2, int relu, void* const* rmask, void* const* y1, void* const* rmask1,
                          void* stream);
int serl_convs_dgrad(int ngroups, const void* const* dz, const void* const* wd, const void* const* xact,
                     void* const* dx, int B, int IH, int IW, int C, int KH, int KW, int stride, int F, int xact_bits,
                     void* stream);
int serl_convs_wgrad(int ngroups, const void* const* x, const void* const* dz, float* const* dw, float* const* dbias,
                     float* workspace, long long workspace_bytes, int in_u8, int B, int IH, int IW, int C, int KH,
                     int KW, int stride, int F, void* stream);
/* conv2's data gradient and conv1's weight + bias gradient in ONE launch (the backward of two_d4pg_encoder.py:30-38's
 * first two layers): conv1 reads the uint8 frames, so dz1 = ReLU'(act1) * dgrad(conv2)(dz2) is needed by conv1's weight
 * gradient only; here it goes from the data gradient's accumulators into the weight-gradient MFMAs and is never written
 * to memory (0.26 GB per 2-camera step at the headline size).  dz2 / wd2 as for serl_convs_dgrad of layer 2 (3x3
 * stride 2, 32 -> 64), rmask1 = the ReLU byte mask serl_convs_fwd wrote for layer 1 (or NULL: no mask), frames uint8
 * [B][IH1][IW1][C1]; dw1 (3,3,C1,32) and dbias1 (32) get the gradient of serl_convs_fwd(in_u8 = 1) (incl. its 1/255).
 * Replaces serl_convs_dgrad + serl_convs_wgrad(in_u8 = 1): the same dz1 bits go into the products, the fp32 sums run in
 * another order.  workspace: serl_convs_dgrad_wgrad1_workspace_bytes(ngroups) bytes are always enough.
 * serl_convs_dgrad_wgrad1_supported returns 1 when the geometry fits (C1 = 9, F1 = 32, F2 = 64, IW1 * C1 a multiple of 16,
 * conv1's output at most 64 wide) and the patch kernels are enabled. */
int serl_convs_dgrad_wgrad1_supported(int IH1, int IW1, int C1, int F1, int F2);
long long serl_convs_dgrad_wgrad1_workspace_bytes(int ngroups);
int serl_convs_dgrad_wgrad1(int ngroups, const void* const* dz2, const void* const* wd2, const void* const* rmask1,
                            const void* const* frames_u8, float* const* dw1, float* const* dbias1, float* workspace,
                            long long workspace_bytes, int B, int IH1, int IW1, int C1, int F1, int F2, void* stream);
/* serl_convs_wgrad without a workspace argument and without any allocation.  NO COUNTERPART IN THE REFERENCE: it is for
 * callers that cannot call hipMalloc -- code under HIP graph capture, tight serving loops, harnesses that forbid it.
 * The split-K slabs live in a 64 MiB file-scope device array of the library (one copy per GPU, held until the process
 * ends); shapes whose slabs need more (serl_conv2d_wgrad_workspace_bytes > serl_convs_wgrad_noalloc_max_bytes) are
 * rejected with status 1.  Results are bit-identical to serl_convs_wgrad (the slab count, i.e. the summation order,
 * depends on the geometry and the device only, never on the size of a workspace).  All calls on one GPU share the
 * array: calls that overlap in time on one GPU (different streams or threads) race.  The first call on a device looks
 * up the array's address (hipGetSymbolAddress) and sets kernel attributes: make one warm-up call per device OUTSIDE
 * any stream capture. */
long long serl_convs_wgrad_noalloc_max_bytes(void);
int serl_convs_wgrad_noalloc(int ngroups, const void* const* x, const void* const* dz, float* const* dw,
                             float* const* dbias, int in_u8, int B, int IH, int IW, int C, int KH, int KW, int stride,
                             int F, void* stream);

/* ---- SpatialLearnedEmbeddings, serl/networks/spatial.py:13-27 (F must be 8) */
int serl_sle_fwd(int ngroups, const void* const* feat, const float* const* ker, float* const* out, int ldo, int B,
                 int HW, int C, int F, int feat_split, void* stream);
int serl_sle_bwd(int ngroups, const float* const* dout, int ldo, const float* const* ker, const void* const* feat,
                 void* const* dfeat, float* const* dker, int B, int HW, int C, int F, int feat_split, void* stream);
/* Grouped like the convolutions (<= 8 groups = cameras x passes per launch); dfeat / dker may be NULL tables. */
/* feat_split bit 0: feat / dfeat are S-format tensors (below) instead of fp32.  serl_sle_bwd folds the ReLU of the last
 * conv layer into dfeat (dfeat = 0 where feat <= 0); feat_split bit 1 (value 2) turns that mask off (features that are
 * not a ReLU output: the frozen-backbone embeddings of the VICE gradient penalty). */

/* ---- frozen MobileNetV3-Large feature extractor (inference), the backbone of TwoMobileNetEncoder / MobileNetEncoder:
 * serl/networks/encoders/two_mobilenet_encoder.py:19-34 ((x - mean) / std, mobilenet.apply(params, x, training=False)),
 * mobilenet_encoder.py:29-48.  The network comes from the un-vendored jeffnet package in the reference
 * ('tf_mobilenetv3_large_100'); this is the published architecture with BatchNorm folded, cut after the last 1x1
 * convolution: features (n, ceil(H/32), ceil(W/32), 960) fp32 NHWC.  params: one flat fp32 buffer of
 * serl_mbv3_param_floats() floats, leaves in forward order, each 64-float aligned (serl_amd/networks/mobilenet.py).
 * images: (n,H,W,3) uint8 (divided by 255 on load) or float already in [0,1]; ImageNet normalisation is applied on load. */
/* 1 (default): the high-resolution blocks run 1x1 expand + depthwise conv as one kernel (the expanded tensor stays in LDS;
 * same bits as the two launches); 0: two launches.  Process-wide, for A/B measurements and the bit-identity test. */
int serl_mbv3_set_fused(int enable);
long long serl_mbv3_param_floats(void);
int serl_mbv3_feature_shape(int H, int W, int* fh, int* fw, int* fc);
long long serl_mbv3_workspace_bytes(int n, int H, int W);
int serl_mbv3_forward(const float* params, const void* images, int in_u8, int n, int H, int W, float* workspace,
                      long long workspace_bytes, float* features, void* stream);

/* ---- batched / segmented dense GEMM: nn.Dense of serl/networks/mlp.py:21-25, pixel_multiplexer.py:34,39, the
 * nn.vmap ensemble of serl/networks/ensemble.py:14-22 (nbatch = members) and their backward products.
 * C[z][m][n] = sum_k A[z][m][k] B[z][k][n] (+bias[z][n]) (act) ; element strides as named; k = seg*seglen + r;
 * relu = activation code of the epilogue: 0 none, 1 ReLU, 2 hard-swish, 3 hard-sigmoid, 4 sigmoid. */
int serl_gemm(const float* a, const float* b, float* c, const float* bias, int nbatch, int M, int N, int K, int seglen,
              long long a_zs, long long a_ss, long long a_ms, long long a_ks, long long b_zs, long long b_ss,
              long long b_ks, long long b_ns, long long c_zs, long long c_ms, long long bias_zs, int nsplit,
              long long c_split_stride, int relu, int accumulate, int a_kcontig, int b_ncontig, void* stream);

/* ---- bias + LayerNorm (flax eps 1e-6) + activation (0 none, 1 relu, 2 tanh, 3 leaky_relu slope 0.01): serl/networks/mlp.py:27-34,
 * pixel_multiplexer.py:35-36,42-43; sums `nslab` split-K partials first. Rows are grouped per ensemble member. */
int serl_ln_act_fwd(const float* pre, int nslab, long long pre_slab_stride, int ldpre, const float* bias,
                    const float* gamma, const float* beta, long long param_group_stride, float* y, int ldy,
                    float* xhat, float* rstd, int R, int N, int rows_per_group, int use_ln, int act, void* stream);
/* ... followed by the Dense(1) head of StateActionValue (serl/networks/state_action_value.py:17-19) on the activated
 * row: q[row] = y[row,:] . head_w[e,:] + head_b[e] (= serl_ln_act_fwd + serl_rowdot in one launch) */
int serl_ln_act_head_fwd(const float* pre, int nslab, long long pre_slab_stride, int ldpre, const float* bias,
                         const float* gamma, const float* beta, long long param_group_stride, float* y, int ldy,
                         float* xhat, float* rstd, int R, int N, int rows_per_group, int use_ln, int act,
                         const float* head_w, long long head_w_group_stride, const float* head_b,
                         long long head_b_group_stride, float* q, void* stream);
/* PixelMultiplexer trunk behind the split-K GEMM of Dense_0 (serl/networks/pixel_multiplexer.py:34-43), one launch:
 * y0 = act(LN(sum of nslab partials + bias0)) (N0 columns) and the state branch y1 = act(LN(x1[:, :K1] W1 + bias1))
 * (N1 columns, the K1-term products in fp32 FMA); y0 / y1 share the row stride ldy (columns of the same z matrix). */
int serl_trunk_ln_fwd(const float* slab, int nslab, long long slab_stride, int N0, const float* bias0,
                      const float* gamma0, const float* beta0, float* y0, float* xhat0, float* rstd0, const float* x1,
                      int ldx1, int K1, const float* W1, int N1, const float* bias1, const float* gamma1,
                      const float* beta1, float* y1, float* xhat1, float* rstd1, int ldy, int R, int act, void* stream);
int serl_ln_act_bwd(const float* dy, int lddy, const float* dq, const float* w_head, long long w_group_stride,
                    const float* y, int ldy, const float* xhat, const float* rstd, const float* gamma,
                    long long param_group_stride, float* dz, float* g, int R, int N, int rows_per_group, int use_ln,
                    int act, void* stream);
/* nseg <= 4 LayerNorm + activation backward problems over the same R rows in one launch, dy = sum of dy_nslab split-K
 * partials (slab stride in floats): the two trunk branches of PixelMultiplexer (pixel_multiplexer.py:34-43) reading
 * the critic's input gradient from the slabs of the GEMM that produced it. Argument tables are host arrays. */
int serl_ln_act_bwd_seg(int nseg, const float* const* dy, const int* lddy, int dy_nslab, long long dy_slab_stride,
                        const float* const* y, const int* ldy, const float* const* xhat, const float* const* rstd,
                        const float* const* gamma, float* const* dz, float* const* g, int R, const int* N, int act,
                        void* stream);
/* column sums over the batch (bias / LayerNorm / Dense(1) gradients), fixed order */
int serl_colsum(const float* x0, const float* w, const float* mul, float* out0, const float* x1, float* out1,
                const float* x2, const float* mul2, float* out2, int ngroups, int rows_per_group, int N, int ldx,
                long long out_group_stride, void* stream);
/* Dense(1) Q head + squeeze, serl/networks/state_action_value.py:17-19 */
int serl_rowdot(const float* h, const float* w, long long w_group_stride, const float* b, long long b_group_stride,
                float* q, int R, int N, int rows_per_group, void* stream);
int serl_copy2d(float* dst, int ldd, const float* src, int lds, int rows, int cols, void* stream); /* jnp.concatenate */
int serl_fill(float* p, float v, long long n, void* stream);

/* nn.Dropout in training mode on a (rows, cols) block with row stride ld, in place (applied to the gradient with the
 * same key in the backward pass): serl/networks/encoders/two_mobilenet_encoder.py:34, encoded_encoder.py:34.
 * keep = jax.random.bernoulli(key, keep_prob, (total_rows, cols))[row_offset + r][c]; row_offset = -1 with total_rows = 1:
 * every row gets row 0 of a (1, cols) draw (the mask under vmap(grad(...)) with a closed-over rng, vice_learner.py:453-460). */
int serl_dropout(float* x, int ld, int rows, int cols, unsigned k0, unsigned k1, float keep_prob, int row_offset,
                 int total_rows, void* stream);

/* VICE reward, serl/agents/vice/vice_learner.py:262-274,301: sigmoids[c][i] = sigmoid(logits[c][i]) (optional outputs),
 * mean_out[i] = mean over the ncam <= 4 cameras, rewards[i] = log(mean / (1 - mean)); argument tables are host arrays. */
int serl_vice_reward(int ncam, const float* const* logits, float* const* sigmoids, float* mean_out, float* rewards, int n,
                     void* stream);

/* ---- VICELearner.update_classifier, serl/agents/vice/vice_learner.py:316-474 (csrc/vice.hip)
 * row_combine: out[r] = wa[r] x[ia[r]] + wb[r] x[ib[r]] over rows of `cols` floats, ia / ib clamped to [0, src_rows)
 *   (mixup :361, GP interpolation :428).
 * bce_mixup: loss[0] = mean_i [lam bce(y_i, ya_i) + (1 - lam) bce(y_i, yb_i)] (optax.sigmoid_binary_cross_entropy, :443-449),
 *   dy = its gradient.  gp_norm: norms[i] = sqrt(sum_j (G_ij^2 + 1e-6)) (:463), alpha[i] = scale 2 (n_i - 1) / (rows n_i).
 * ln_tangent_fwd / _bwd: forward tangent of [LayerNorm] + activation and its adjoint w.r.t. the tangent AND the primal
 *   input -- the two pieces of the hand-derived double backward of the gradient penalty (:453-467); acts as serl_ln_act_*. */
int serl_row_combine(const float* x, int src_rows, const int* ia, const int* ib, const float* wa, const float* wb, float* out,
                     int rows, long long cols, void* stream);
int serl_bce_mixup(const float* y, const float* ya, const float* yb, float lam, int n, float* dy, float* loss, void* stream);
int serl_gp_norm(const float* G, int rows, long long D, float scale, float* alpha, float* norms, void* stream);
int serl_ln_tangent_fwd(const float* a, const float* xhat, const float* rstd, const float* gamma, const float* y, float* out,
                        float* w, int R, int N, int use_ln, int act, void* stream);
int serl_ln_tangent_bwd(const float* outbar, const float* tbar_ext, const float* y, const float* w, const float* a,
                        const float* xhat, const float* rstd, const float* gamma, float* abar, float* ubar, float* ggamma,
                        float* nbar, int R, int N, int use_ln, int act, void* stream);
int serl_axpy(float* y, const float* x, float a, long long n, void* stream);

/* ---- tanh-Gaussian policy head, serl/distributions/tanh_normal.py:29-48 + TFP MultivariateNormalDiag/Tanh;
 * eps = jax.random.normal(key, (total_rows, A))[row_offset : row_offset + B] (threefry2x32, XLA erf_inv). */
int serl_tanh_gaussian_fwd(const float* mean, const float* log_std_raw, int ld_ms, float* act, int ld_act, float* xpre,
                           float* eps, float* logp, int B, int A, unsigned k0, unsigned k1, int row_offset,
                           int total_rows, int deterministic, void* stream);
int serl_tanh_gaussian_bwd(const float* act, int ld_act, const float* eps, const float* log_std_raw, int ld_ms,
                           const float* da, int ld_da, const float* log_temp, float inv_batch, float* dmean,
                           float* dlstd, int ld_out, int B, int A, void* stream);
/* log pi(a) of GIVEN actions (B, A): TanhTransformedDistribution.log_prob (serl/distributions/tanh_transformed.py:17-30)
 * with the Tanh bijector's float32 inverse atanh(clip(a, +-nextafter(1, 0))); logp (B). */
int serl_tanh_gaussian_log_prob(const float* mean, const float* log_std_raw, int ld_ms, const float* actions, int ld_act,
                                float* logp, int B, int A, void* stream);
/* Ensemble(StateActionValue(MLP)) forward with hidden dims (256, 256) in ONE launch (csrc/ensemble.hip; serl/networks/
 * ensemble.py:8-22, state_action_value.py:11-19, mlp.py:18-35): Dense -> [LayerNorm] -> ReLU twice, then the Dense(1) head.
 * Member e's leaves sit e * stride floats behind the given pointers (a negative stride walks a REDQ subsample backwards).
 * h0 / h1 (post-ReLU activations), xh0 / xh1 / rs0 / rs1 (LayerNorm x-hat and 1/sigma) are what the backward reads; all
 * six may be NULL for passes nobody differentiates.  Bit-identical to serl_gemm + serl_ln_act_fwd + serl_ln_act_head_fwd
 * (or serl_rowdot); precision "bf16_3x" only. */
int serl_ensemble_fwd_fused(const float* inp, int ld_inp, int din, int B, int E, const float* w0, const float* b0,
                            const float* g0, const float* be0, const float* w1, const float* b1, const float* g1,
                            const float* be1, const float* hw, const float* hb, long long w0_member_stride,
                            long long w1_member_stride, long long vec_member_stride, long long head_bias_member_stride,
                            int use_ln, float* h0, float* h1, float* xh0, float* xh1, float* rs0, float* rs1, float* q,
                            void* stream);
/* Two serl_ensemble_fwd_fused problems of the same batch size and input width in ONE launch -- the state-based learners'
 * target pass (REDQ-subsampled members, next observations) and critic pass (all members, observations) of
 * sac_learner.py:177-206, both on the caller's stream.  Fields as the arguments of serl_ensemble_fwd_fused; bit-identical
 * to two calls of it. */
typedef struct {
  const float* inp; int ld_inp, din, B, E;
  const float *w0, *b0, *g0, *be0, *w1, *b1, *g1, *be1, *hw, *hb;
  long long w0_member_stride, w1_member_stride, vec_member_stride, head_bias_member_stride;
  int use_ln;
  float *h0, *h1, *xh0, *xh1, *rs0, *rs1, *q;
} serl_ens_fwd_job;
int serl_ensemble_fwd_fused_pair(const serl_ens_fwd_job* a, const serl_ens_fwd_job* b, void* stream);
/* TanhNormal(MLP(256, 256)) policy forward in ONE launch (csrc/ensemble.hip): Dense + ReLU twice, the two Dense heads,
 * log-std clip, the jax.random.normal draw keyed by the global element index, a = tanh(mean + sigma eps) and log pi(a)
 * (serl/distributions/tanh_normal.py:17-53 behind serl/networks/mlp.py:18-35; sampled at sac_learner.py:126,169 and
 * serl/agents/agent.py:12-22) -- what three serl_gemm launches + serl_tanh_gaussian_fwd compute.  z (B, ld_z): the first
 * din columns are the MLP input.  h0 / h1 (B, 256) post-ReLU activations, ms (B, 2A) raw [mean | log_std], eps (B, A),
 * logp (B) may be NULL.  action_dim <= 8; precision "bf16_3x" only. */
int serl_policy_fwd_fused(const float* z, int ld_z, int din, int B, const float* w0, const float* b0, const float* w1,
                          const float* b1, const float* mk, const float* lk, const float* mb, const float* lb, int A,
                          unsigned k0, unsigned k1, int row_offset, int total_rows, int deterministic, float* act,
                          int ld_act, float* ms, float* eps, float* logp, float* h0, float* h1, void* stream);
/* Backward of the same network for all E members of the critic (member e's leaves e members behind the pointers) from
 * dq (E, B) = dLoss/dq, in ONE launch (csrc/ensemble.hip): dz1 / dz0 = gradients w.r.t. the outputs of the second / first
 * Dense, g1 / g0 = gradients w.r.t. the LayerNorm outputs (use_ln only; NULL otherwise) -- what serl_ln_act_bwd (rank-1
 * head form) + serl_gemm + serl_ln_act_bwd compute in three launches (sac_learner.py:200-211 differentiates
 * ensemble.py:8-22).  serl_ensemble_colsum: every batch-column sum of that backward in one launch -- head kernel / bias
 * gradient, LayerNorm scale / bias gradients (use_ln), Dense bias gradients; (E, 256) each, d_hb (E); with inp (B, ld_inp)
 * and d_w0 (E, din, 256) non-NULL (din <= 32: the state-based learners) also the first Dense's kernel gradient
 * d_w0[e] = inp^T dz0[e], in exact fp32.  serl_ensemble_bwd_fused: "bf16_3x" only. */
int serl_ensemble_bwd_fused(const float* dq, int B, int E, const float* hw, const float* w1, const float* g0s,
                            const float* g1s, int use_ln, const float* h0, const float* h1, const float* xh0,
                            const float* xh1, const float* rs0, const float* rs1, float* dz1, float* g1, float* dz0,
                            float* g0, void* stream);
int serl_ensemble_colsum(const float* dq, const float* h1, const float* g1, const float* xh1, const float* dz1,
                         const float* g0, const float* xh0, const float* dz0, int B, int E, int use_ln, float* d_hw,
                         float* d_hb, float* d_lnb1, float* d_lns1, float* d_b1, float* d_lnb0, float* d_lns0, float* d_b0,
                         const float* inp, int ld_inp, int din, float* d_w0, void* stream);
/* behaviour-cloning losses: -mean log_prob of GIVEN actions (clipped to +-(1 - 1e-5), x = atanh(a)) under the tanh-Gaussian
 * policy with its gradient wrt mean / raw log-std (serl/agents/bc/pixel_bc_learner.py:114-134), info[0] = sum of the
 * per-row nll; sigmoid BCE of logits against (action column > 0) (pixel_hybrid_bc_learner.py:167-176), info[0] = sum. */
int serl_tanh_gaussian_nll(const float* mean, const float* log_std_raw, int ld_ms, const float* actions, int ld_act, int B,
                           int A, float inv_batch, float* dmean, float* dlstd, int ld_out, float* info, void* stream);
int serl_bce_logits(const float* logits, const float* action_column, int ld_act, int B, float inv_batch, float* dlogit,
                    float* info, void* stream);

/* ---- TD target (ensemble min, optional entropy backup) + critic loss gradient, serl/agents/sac/sac_learner.py:185-209 */
int serl_td_loss(const float* next_q, int Emin, const float* rew, const float* mask, const float* next_logp,
                 const float* log_temp, float discount, const float* q, int E, int B, float inv_count, float* target,
                 float* dq, float* info, void* stream);
/* actor loss sums, sac_learner.py:135-139 */
int serl_actor_loss(const float* q, int E, int B, const float* logp, const float* log_temp, float* info, void* stream);
/* optax.adam + optax.incremental_update fused over a flat buffer, sac_learner.py:142,212-217 */
int serl_adam_polyak(float* p, float* m, float* v, const float* g, float* target, long long n, float lr, float b1,
                     float b2, float eps, float bc1, float bc2, float tau, float grad_scale, void* stream);
/* temperature loss + scalar Adam, sac_learner.py:146-158, sac/temperature.py:9-14 */
int serl_temperature_update(float* state, const float* sum_logp, float inv_batch, float target_entropy, float lr,
                            float b1, float b2, float eps, float bc1, float bc2, float* info_out, void* stream);

/* ---- replay gather + _unpack + DrQ random shift, uint8, bit-exact:
 * serl/data/memory_efficient_replay_buffer.py:155-167, serl/utils/commons.py:16-24, serl/utils/augmentations.py:5-11,44-48.
 * out (n,H,W,3*S)[.., c*S+s] = frame (t0+s) of sample j's T-frame window, shifted by off[j] - pad with edge clamp. */
int serl_gather_crop_frames(const unsigned char* frames, long long cap, const long long* indx, int T, int t0,
                            const int* off, unsigned char* out, int n, int H, int W, int S, int pad, void* stream);
int serl_gather_crop_packed(const unsigned char* src, int T, int t0, const int* off, unsigned char* out, int n, int H,
                            int W, int S, int pad, void* stream);
int serl_gather_window(const unsigned char* frames, long long cap, const long long* indx, unsigned char* out, int n,
                       int HW3, int S, void* stream);
/* (dy,dx) = jax.random.randint(split(key, n_total)[row], (2,), 0, 2*pad+1), augmentations.py:6,46 */
int serl_crop_offsets(unsigned k0, unsigned k1, int n, int row_offset, int n_total, int pad, int* out, void* stream);
/* device-resident sampling, Dataset.sample_jax serl/data/dataset.py:106-129 (GPUReplayBuffer.get_iterator,
 * serl/data/replay_buffer.py:149-152): out (n,) int32 = jax.random.randint(key, (n,), minval, maxval) in device
 * memory, *out_max (device, may be NULL) = out.max(); take_rows = jnp.take(src, indx, axis=0) on rows of row_bytes. */
int serl_prng_randint(unsigned k0, unsigned k1, int n, int minval, int maxval, int* out, int* out_max, void* stream);
int serl_take_rows(const void* src, long long nrows, long long row_bytes, const int* indx, void* out, int n,
                   void* stream);

/* ---- host-side jax.random key plumbing (no GPU needed): split / bits / choice(replace=False),
 * sac_learner.py:64,121-122,168-198; serl/networks/ensemble.py:28 */
int serl_prng_split(unsigned k0, unsigned k1, int num, unsigned* out);
int serl_prng_bits(unsigned k0, unsigned k1, int n, unsigned* out);
int serl_prng_choice(unsigned k0, unsigned k1, int n, int m, int* out);
int serl_prng_fold_in(unsigned k0, unsigned k1, unsigned data, unsigned* out);

/* ==== Native step driver: one call per update() / per action (serl_amd/csrc/driver.hip).
 * The coarse boundary of SURVEY.md 8(b): Python holds handles and pointers, every launch, stream, event and per-step
 * scalar (jax split chain, REDQ member strides, Adam bias corrections) of
 *   SACLearner.update    serl/agents/sac/sac_learner.py:221-238 (update_critic :160-219, update_actor :120-144,
 *                        update_temperature :146-158)
 *   DrQLearner.update    serl/agents/drq/drq_learner.py:154-186 (+ _unpack / random crop, utils/commons.py:7-29,
 *                        utils/augmentations.py:5-11,44-48) and PixelDDPGLearner.update (ddpg/pixel_ddpg_learner.py:148-181)
 *   Agent.sample_actions / eval_actions   serl/agents/agent.py:12-37
 * is produced in C.  serl_learner_create allocates host state, HIP streams and events; device workspace buffers
 * (activations etc.) are allocated by the handle on first use of a minibatch size and freed by serl_learner_destroy.
 * One handle serves one learner thread (update) and any number of acting threads (act, serialised internally). */
typedef struct serl_learner serl_learner;
typedef struct {
  int action_dim, num_hidden, hidden_dims[4], num_qs;
  int num_min_qs;              /* 0 = None (no REDQ subsample) */
  int critic_layer_norm, backup_entropy;
  int has_temperature;         /* 0: DDPG variants (no entropy term: temp[0] = -inf, no temperature step) */
  int pixel, obs_dim;          /* state-based learners: pixel = 0, observations (n, obs_dim) */
  int num_cameras, height, width, stack, state_dim, latent_dim, num_conv;
  int cnn_features[8], cnn_filters[8], cnn_strides[8];
  int aug_rank[4];             /* position of the i-th camera (sorted-key order) in the caller's pixel_keys order */
  float discount, tau, target_entropy, actor_lr, critic_lr, temp_lr;
  int encoder;                 /* 0 TwoD4PGEncoder (trainable conv stack), 1 TwoMobileNetEncoder (frozen backbone, stack 1) */
  int feat_h, feat_w, feat_c;  /* encoder 1: serl_mbv3_feature_shape(height, width) */
  float encoder_dropout;       /* encoder 1: Dropout rate behind the spatial embeddings in training passes (0.1) */
  int critic_no_action;        /* != 0: the "critic" slot is the hybrid BC learner's discrete actor: MLP on the latent only */
  float critic_dropout;        /* critic_dropout_rate (sac_learner.py:47, mlp.py:26-29): nn.Dropout behind every hidden Dense of
                                  the critic MLPs (before LayerNorm) in the training=True applies -- target critic, critic,
                                  the actor step's critic; 0 = None.  Mask stream BUILD-DEFINED (flax folds the module path and
                                  the vmap axis into the rng): bernoulli(fold_in(fold_in(key, layer), member position), keep)
                                  over the (global minibatch, width) array, keyed by the global row */
} serl_learner_config;
/* flat fp32 device buffers of ONE agent copy (parameters, Adam moments, temperature state {log_temp, mu, nu}) plus the
 * gradient buffers (same layouts); update() steps them in place -- the caller clones first to keep the old agent valid */
typedef struct {
  float *actor, *actor_mu, *actor_nu, *critic, *critic_mu, *critic_nu, *target, *temp, *critic_grads, *actor_grads;
  int actor_step, critic_step, temp_step;   /* optimizer step counts BEFORE this update */
} serl_agent_buffers;
typedef struct {
  int kind;                          /* 0 device frame store + window indices, 1 packed (n,H,W,3,S+1), 2 unpacked */
  const unsigned char* frames; long long capacity; const long long* indx;   /* kind 0 */
  const unsigned char* obs;          /* kind 1: the packed tensor; kind 2: observations (n,H,W,3,S) */
  const unsigned char* next_obs;     /* kind 2: next_observations (n,H,W,3,S) */
} serl_pixel_source;
typedef struct {
  int n;                             /* rows = minibatch * utd_ratio (this rank's rows) */
  int aug_mode;                      /* 0 state-based, 1 DrQLearner key order, 2 PixelDDPGLearner key order */
  const float *actions, *rewards, *masks;
  const float *obs, *next_obs;       /* state-based */
  const float *state, *next_state;   /* pixel-based */
  serl_pixel_source pixels[4];       /* cameras in sorted-key order */
  int features_hook;                 /* != 0: call the hook with which = 3 once the features of all rows are enqueued */
  int use_sac_rng; unsigned sac_rng[2];   /* != 0: the SAC update's split chain starts from sac_rng instead of the key the
                                        augmentation left (VICELearner.update: vice_reward advances it, :255-260,275) */
} serl_update_batch;
/* data-parallel hook: sum buffer `which` over the ranks (0 critic_grads, 1 actor_grads, 2 the first 4 info floats --
 * serl_learner_update passes info_out for those), behind the work enqueued on the update's stream; returns 0 on success.
 * which = 3 ("features ready", only when serl_update_batch.features_hook != 0): the augmented images and, for the
 * MobileNet encoder, the backbone features of all rows are enqueued; the hook may enqueue work on the update's stream
 * that overwrites the batch's rewards before the first critic step reads them (VICELearner.update, vice_learner.py:300-308) */
typedef int (*serl_allreduce_fn)(void* ctx, int which);
/* Pixel learners with world > 1 split buffer 0 in two buckets (SERL_AMD_DP_OVERLAP=0: one all-reduce as above):
 * which = 4: the SUFFIX of critic_grads behind the image_encoder leaves (trunk Dense / LayerNorms / ensemble: 17 of 21 MB
 * at the headline config), called as soon as the kernels that write it are enqueued -- the hook must enqueue this
 * all-reduce on the stream serl_learner_comm_stream returns, where it runs next to the encoder backward; which = 5: the
 * image_encoder PREFIX, on the update's stream behind the last weight gradient.  The driver orders both against the
 * optimizer step with events; no host synchronisation anywhere (the contract of an RCCL all-reduce).  No counterpart in
 * the reference (single device; its commented-out batch split: serl/data/replay_buffer.py:85-98). */
/* critic_offsets / actor_offsets: offsets in floats of every leaf of the flat stores in the layout order of
 * serl_amd/networks/spec.py (encoder + trunk leaves first for pixel learners), followed by the store's total size */
int serl_learner_create(const serl_learner_config* cfg, const long long* critic_offsets, int n_critic,
                        const long long* actor_offsets, int n_actor, serl_learner** out);
int serl_learner_destroy(serl_learner* h);
/* encoder 1: the frozen backbone's parameters (device, serl_mbv3_param_floats floats); must outlive the handle's use */
int serl_learner_set_backbone(serl_learner* h, const float* params);
/* rng_in / rng_out: jax key (uint32[2]) before / after the call; info_out: 8 device floats {critic loss sum, q sum,
 * actor loss sum, logp sum, temperature, temperature loss, entropy, -}; everything is enqueued on `stream` and the
 * handle's own streams, nothing waits for the GPU */
int serl_learner_update(serl_learner* h, const serl_agent_buffers* agent, const serl_update_batch* batch, int utd_ratio,
                        const unsigned* rng_in, unsigned* rng_out, float* info_out, void* stream,
                        serl_allreduce_fn allreduce, void* allreduce_ctx, int rank, int world);
/* images: per camera (sorted order) a device (H,W,3*S) uint8 image; state_or_obs: device state / observation vector;
 * key NULL or deterministic != 0: mode of the policy; act_out / eps_out (may be NULL): device, action_dim floats */
int serl_learner_act(serl_learner* h, const float* actor_params, const unsigned char* const* images,
                     const float* state_or_obs, const unsigned* key, int deterministic, float* act_out, float* eps_out,
                     void* stream);
/* ---- behaviour cloning: PixelBCLearner.update (serl/agents/bc/pixel_bc_learner.py:104-139) and
 * PixelHybridBCLearner.update / compute_actions (pixel_hybrid_bc_learner.py:123-185).  actor: the TanhNormal policy (actor
 * layout, its encoder is trained); disc (NULL for PixelBCLearner): the discrete actor in the handle's critic layout
 * (num_qs = 1, no LayerNorm, critic_no_action = 1).  batch: n, actions (n, ld_actions), state, pixels (observations only;
 * kind 2 reads .obs).  key_aug / key1 / key2: the augmentation key and the two dropout keys of the call (jax keys, host).
 * info_out: 2 device floats {sum of per-row nll, sum of per-row BCE}. */
typedef struct {
  float *actor, *actor_mu, *actor_nu, *actor_grads, *disc, *disc_mu, *disc_nu, *disc_grads;
  int actor_step, disc_step;        /* optimizer step counts BEFORE this update */
} serl_bc_buffers;
int serl_learner_bc_update(serl_learner* h, const serl_bc_buffers* bc, const serl_update_batch* batch, int ld_actions,
                           const unsigned* key_aug, const unsigned* key1, const unsigned* key2, float* info_out,
                           void* stream);
int serl_learner_bc_logit(serl_learner* h, const float* disc_params, const unsigned char* const* images,
                          const float* state, float* logit_out, void* stream);
/* test / tooling access: copy a named workspace buffer of the update at minibatch size B ("aug_<0 obs|1 next>_<cam>") */
int serl_learner_read_buffer(serl_learner* h, int B, const char* name, void* dst, long long nbytes, void* stream);
/* device address and size of a named workspace buffer ("bb_<side>_<cam>": the frozen backbone's features of the augmented
 * images, (n, fh, fw, fc) fp32) -- what a `features ready` hook (serl_allreduce_fn, which = 3) reads */
int serl_learner_buffer_ptr(serl_learner* h, int B, const char* name, void** ptr, long long* nbytes);
int serl_learner_comm_stream(serl_learner* h, void** stream);
long long serl_learner_workspace_bytes(serl_learner* h);
/* bench.py instrumentation: HIP events around every conv launch (on its launch stream) / between the phases of a step */
int serl_learner_set_timing(serl_learner* h, int conv_events, int phase_events);
int serl_learner_timing_count(serl_learner* h);
int serl_learner_timing_get(serl_learner* h, int i, int* desc, double* val);
int serl_learner_phase_count(serl_learner* h);
int serl_learner_phase_get(serl_learner* h, int i, const char** name, double* ms_to_next);
int serl_learner_timing_reset(serl_learner* h);

#ifdef __cplusplus
}
#endif
#endif
