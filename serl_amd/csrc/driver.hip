// Native step driver: ONE C-ABI call enqueues a whole `update(batch, utd_ratio)` of the SAC / DrQ learners
// (serl/agents/sac/sac_learner.py:221-238 with update_critic :160-219, update_actor :120-144, update_temperature
// :146-158; serl/agents/drq/drq_learner.py:154-186) -- every kernel launch, all streams and events, the jax split chain
// of the call, REDQ member strides and Adam bias corrections are computed here; Python passes pointers and gets the
// new rng back.  The same for one `sample_actions` / `eval_actions` (serl/agents/agent.py:12-37).
//
// The launch sequence (what runs on which stream) is DESIGN.md section 3b:
//   main   : gather+crop of all B*U rows, then per critic step: target + critic encoders (grouped launches), target
//            trunk + ensemble, TD loss, critic backward (data-gradient chain), Adam + Polyak
//   side 1 : critic trunk + ensemble forward of the step (next to the target chain)
//   gside  : parameter-gradient kernels of the backward (column sums, dW GEMMs, conv weight gradients)
//   actor  : the actor's no-grad pass over next_obs for step i+1 (its parameters are constant during the UTD loop),
//            gated to start when main leaves its HBM-bound encoder forward; behind the last one, the actor's own pass
//            over obs for the actor step
// Host cost per launch is a plain function call + hipLaunchKernel (~3-5 us) instead of ~21 us of Python per launch.
#include <math.h>

#include <algorithm>
#include <functional>
#include <mutex>
#include <string>
#include <unordered_map>
#include <stdlib.h>
#include <vector>

#include "../../include/serl_hip.h"
#include "common.h"

#define DRV(expr)            \
  do {                       \
    int e__ = (expr);        \
    if (e__) return e__;     \
  } while (0)
#define DRV_HIP(expr) SERL_CHECK_HIP(expr, "learner")

#include <chrono>
namespace {

// SERL_AMD_HOST_PROF=1: host time of the driver by category (event record / wait calls, workspace lookups, whole update
// calls), printed when a handle is destroyed -- where the ~10 ms of CPU time per pixel update go
struct HostProf {
  bool on = getenv("SERL_AMD_HOST_PROF") != nullptr && atoi(getenv("SERL_AMD_HOST_PROF")) != 0;
  double t_rec = 0, t_wait = 0, t_ws = 0, t_upd = 0;
  long n_rec = 0, n_wait = 0, n_ws = 0, n_upd = 0;
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};
HostProf g_hp;
// Cross-stream dependencies.  hipStreamWaitEvent costs ~15-22 us of HOST time per call on this ROCm (tools/wait_cost.py;
// 222 of them per pixel update = 4.7-5.3 of its ~10 ms of CPU time), a stream memory operation ~3 us stand-alone.
// OPTION (SERL_AMD_VALUE_WAITS=1, off by default): every event of the driver carries an 8-byte signal word
// (hipMallocSignalMemory), "record" = hipStreamWriteValue32 of the next sequence number behind the work enqueued so far,
// "wait" = hipStreamWaitValue32(>= the number of the latest record) -- the semantics of hipEventRecord /
// hipStreamWaitEvent (a wait refers to the most recent record at the time of the call; the producer's write is always
// enqueued before the consumer's wait, so no wait can be left without its write).  Measured inside the update (round 3,
// two interleaved runs each): waits 4.7 -> 0.5 ms of host time per update, but the write calls cost 3.2 ms behind queued
// kernels (0.3 ms as event records) and the GPU side is SLOWER -- 34.8 vs 32.6 ms per update (the command processor
// polls the word instead of being signalled) -- so the events stay.  All GPU tests pass in either mode.
struct EvSig { uint32_t* word = nullptr; uint32_t last = 0; bool tried = false; };
std::unordered_map<hipEvent_t, EvSig> g_sigs;
std::mutex g_sig_mu;
const bool g_value_waits = getenv("SERL_AMD_VALUE_WAITS") != nullptr && atoi(getenv("SERL_AMD_VALUE_WAITS")) != 0;
inline EvSig* ev_sig(hipEvent_t e) {      // call with g_sig_mu held
  EvSig& g = g_sigs[e];
  if (!g.tried) {
    g.tried = true;
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) == hipSuccess && p) {
      if (hipMemset(p, 0, 8) == hipSuccess) g.word = (uint32_t*)p;
      else (void)hipFree(p);
    } else {
      (void)hipGetLastError();
    }
  }
  return g.word ? &g : nullptr;
}
inline hipError_t ev_record(hipEvent_t e, hipStream_t s) {
  const double t0 = g_hp.on ? HostProf::now() : 0;
  hipError_t r;
  EvSig* g = nullptr;
  uint32_t val = 0;
  if (g_value_waits) {
    std::lock_guard<std::mutex> lock(g_sig_mu);
    g = ev_sig(e);
    if (g) val = ++g->last;
  }
  r = g ? hipStreamWriteValue32(s, g->word, val, 0) : hipEventRecord(e, s);
  if (g_hp.on) { g_hp.t_rec += HostProf::now() - t0; ++g_hp.n_rec; }
  return r;
}
inline hipError_t ev_wait(hipStream_t s, hipEvent_t e) {
  const double t0 = g_hp.on ? HostProf::now() : 0;
  hipError_t r;
  uint32_t* word = nullptr;
  uint32_t val = 0;
  if (g_value_waits) {
    std::lock_guard<std::mutex> lock(g_sig_mu);
    EvSig* g = ev_sig(e);
    if (g) { word = g->word; val = g->last; }
  }
  if (word) r = val == 0 ? hipSuccess : hipStreamWaitValue32(s, word, val, hipStreamWaitValueGte, 0xFFFFFFFFu);   // (never recorded: nothing to wait for)
  else r = hipStreamWaitEvent(s, e, 0);
  if (g_hp.on) { g_hp.t_wait += HostProf::now() - t0; ++g_hp.n_wait; }
  return r;
}
inline void ev_destroy(hipEvent_t e) {
  if (!e) return;
  {
    std::lock_guard<std::mutex> lock(g_sig_mu);
    auto it = g_sigs.find(e);
    if (it != g_sigs.end()) {
      if (it->second.word) (void)hipFree(it->second.word);
      g_sigs.erase(it);
    }
  }
  (void)hipEventDestroy(e);
}

constexpr int TRUNK_DIM = 512;    // hard-coded in serl/networks/pixel_multiplexer.py:34
constexpr int SLE_FEATURES = 8;   // two_d4pg_encoder.py:40
constexpr int MAX_CAM = 4, MAX_CONV = 8, MAX_HID = 4, MAXG_ = 8;
constexpr float ADAM_B1 = 0.9f, ADAM_B2 = 0.999f, ADAM_EPS = 1e-8f;   // optax.adam defaults
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

struct ConvL { int IH, IW, C, K, S, F, OH, OW; };

struct Cfg {
  int A, nh, hid[MAX_HID], E, num_min_qs, ln, backup_entropy, has_temp, pixel;
  int obs_dim;                                   // state-based
  int ncam, H, W, stack, state_dim, latent_dim;  // pixel-based
  int nl; ConvL L[MAX_CONV];
  int encoder;             // 0 TwoD4PGEncoder (trainable conv stack), 1 TwoMobileNetEncoder (frozen backbone + SLE + Dropout)
  int fh, fw, fc;          // feature map the spatial embeddings pool (mobilenet: backbone output; d4pg: last conv layer)
  float enc_drop;          // Dropout rate behind the spatial embeddings (mobilenet encoder, training passes only)
  float crit_drop;         // critic_dropout_rate: Dropout behind every hidden Dense of the critic MLPs (training applies)
  int aug_rank[MAX_CAM];   // position of (sorted) camera ci in the caller's pixel_keys order: augmentation key order
  float discount, tau, target_entropy, actor_lr, critic_lr, temp_lr;
  int enc_dim, latent, ldz;
  int crit_A;               // action columns of the critic's input (0: the BC learners' discrete head sees the latent only)
};

// offsets (in floats) of the leaves of a flat parameter store, in the order serl_amd/networks/spec.py adds them
struct EncOff { long long ck[MAX_CAM][MAX_CONV], cb[MAX_CAM][MAX_CONV], sle[MAX_CAM], d0k, d0b, l0s, l0b, d1k, d1b, l1s, l1b; };
struct CriticOff { EncOff e; long long w[MAX_HID], b[MAX_HID], lns[MAX_HID], lnb[MAX_HID], hk, hb; long long size; };
struct ActorOff { EncOff e; long long w[MAX_HID], b[MAX_HID], mk, lk, mb, lb; long long size; };

struct Store {   // a flat fp32 parameter (or gradient) buffer
  float* base;
  float* at(long long off) const { return base + off; }
};

// ---------------------------------------------------------------- named, lazily allocated device buffers
struct Workspace {
  struct Buf { void* p; size_t bytes; std::string name; };
  // keyed by a 64-bit FNV-1a hash of the name (the name itself is kept and compared on a hit): the ~250 lookups of a
  // critic step no longer build a std::string each
  std::unordered_map<uint64_t, Buf> m;
  char name[96];

  static uint64_t hash_name(const char* s) {
    uint64_t h = 1469598103934665603ull;
    for (; *s; ++s) { h ^= (unsigned char)*s; h *= 1099511628211ull; }
    return h;
  }
  const Buf* find(const char* nm) const {
    auto it = m.find(hash_name(nm));
    return (it != m.end() && it->second.name == nm) ? &it->second : nullptr;
  }
  int get(size_t bytes, void** out, hipStream_t st) {
    ++g_hp.n_ws;
    const uint64_t key = hash_name(name);
    auto it = m.find(key);
    if (it != m.end()) SERL_CHECK_ARG(strcmp(it->second.name.c_str(), name) == 0, "workspace: name hash collision (%s / %s)", name, it->second.name.c_str());
    if (it != m.end() && it->second.bytes == bytes) { *out = it->second.p; return SERL_OK; }
    if (it != m.end()) {   // shape of a name changed: rare (another batch size reuses the handle's act workspace)
      DRV_HIP(hipDeviceSynchronize());
      DRV_HIP(hipFree(it->second.p));
      m.erase(it);
    }
    void* p = nullptr;
    DRV_HIP(hipMalloc(&p, bytes ? bytes : 4));
    // zero fill; the buffer may be used on any of the driver's streams: wait for the fill once, on first use of a name
    DRV_HIP(hipMemsetAsync(p, 0, bytes ? bytes : 4, st));
    DRV_HIP(hipStreamSynchronize(st));
    m[key] = Buf{p, bytes, name};
    *out = p;
    return SERL_OK;
  }
  void release() {
    for (auto& kv : m) (void)hipFree(kv.second.p);
    m.clear();
  }
};
#define WSF(ws, st, var, nfloats, ...)                                   \
  float* var = nullptr;                                                  \
  {                                                                      \
    snprintf((ws).name, sizeof((ws).name), __VA_ARGS__);                 \
    DRV((ws).get((size_t)(nfloats) * 4, (void**)&var, st));              \
  }
#define WSB(ws, st, type, var, nbytes, ...)                              \
  type* var = nullptr;                                                   \
  {                                                                      \
    snprintf((ws).name, sizeof((ws).name), __VA_ARGS__);                 \
    DRV((ws).get((size_t)(nbytes), (void**)&var, st));                   \
  }

// ---------------------------------------------------------------- optional HIP-event instrumentation (bench.py)
struct TimeRec { int kind, in_u8, groups, B, IH, C, F; double flops, bytes; hipEvent_t a, b; };
struct PhaseRec { const char* name; hipEvent_t ev; };

struct GradSide {   // side stream for parameter-gradient kernels: fork() = side waits for main, join() = main waits
  hipStream_t main = nullptr, side = nullptr, own = nullptr;   // own: the handle's side stream; side: the one in use
  std::vector<hipEvent_t> ev;
  size_t n = 0;
  int next(hipEvent_t* e) {
    if (n == ev.size()) {
      hipEvent_t x;
      DRV_HIP(hipEventCreateWithFlags(&x, hipEventDisableTiming));
      ev.push_back(x);
    }
    *e = ev[n++];
    return SERL_OK;
  }
  void begin(hipStream_t m, bool use_side) { main = m; side = use_side ? own : m; n = 0; }
  int fork(hipStream_t* out) {
    if (side == main) { *out = main; return SERL_OK; }
    hipEvent_t e;
    DRV(next(&e));
    DRV_HIP(ev_record(e, main));
    DRV_HIP(ev_wait(side, e));
    *out = side;
    return SERL_OK;
  }
  int join() {
    if (side == main) return SERL_OK;
    hipEvent_t e;
    DRV(next(&e));
    DRV_HIP(ev_record(e, side));
    DRV_HIP(ev_wait(main, e));
    return SERL_OK;
  }
};

}  // namespace

struct serl_learner {
  Cfg cfg;
  CriticOff co;
  ActorOff ao;
  int device = 0;
  std::unordered_map<int, Workspace> ws_update;   // per minibatch size
  Workspace ws_act;
  std::mutex act_mu;
  hipStream_t s1 = nullptr, astream = nullptr;
  // data parallel: the all-reduce of the critic gradient's trunk + ensemble leaves (17 of the 21 MB) is enqueued on this
  // stream as soon as their gradient kernels are, i.e. next to the whole encoder backward; created on first use
  hipStream_t comm = nullptr;
  hipEvent_t comm_fork = nullptr, comm_done = nullptr;
  bool dp_overlap = getenv("SERL_AMD_DP_OVERLAP") == nullptr || strcmp(getenv("SERL_AMD_DP_OVERLAP"), "0") != 0;
  GradSide gside, gside2;   // gside2 (experiments, default off): every other conv weight gradient (SERL_AMD_WG_STREAMS=2), or the heads' parameter gradients (=3)
  // CU cap of the run-ahead pass's persistent patch kernels (default 7/8 of the CUs, set at creation; 0: none).  Their
  // blocks otherwise hold every CU for ~100 us at a time and the short kernels of the critic step's latency-bound chain
  // wait for one to retire: 592.9 -> 596.5 grad-steps/s at 224 of 256 (three interleaved runs; 240: -0.3 %, 208: +0.2 %).
  int actor_cus = getenv("SERL_AMD_ACTOR_CUS") ? atoi(getenv("SERL_AMD_ACTOR_CUS")) : -1;
  int wg_streams = getenv("SERL_AMD_WG_STREAMS") ? atoi(getenv("SERL_AMD_WG_STREAMS")) : 1;
  hipEvent_t fork_ev = nullptr, join1 = nullptr, astart = nullptr, agate = nullptr, aobs = nullptr;
  std::vector<hipEvent_t> adone;
  const float* backbone = nullptr;   // frozen MobileNetV3 parameters (serl_learner_set_backbone)
  // instrumentation
  bool timing = false, phases = false;
  int wg_main_layers = getenv("SERL_AMD_WG_MAIN_LAYERS") ? atoi(getenv("SERL_AMD_WG_MAIN_LAYERS")) : 1;   // A/B switch
  bool ens_fused = getenv("SERL_AMD_ENS_FUSED") == nullptr || strcmp(getenv("SERL_AMD_ENS_FUSED"), "0") != 0;   // A/B switch
  bool pol_fused = getenv("SERL_AMD_POL_FUSED") == nullptr || strcmp(getenv("SERL_AMD_POL_FUSED"), "0") != 0;   // A/B switch
  bool ens_bwd_fused = getenv("SERL_AMD_ENS_BWD_FUSED") == nullptr || strcmp(getenv("SERL_AMD_ENS_BWD_FUSED"), "0") != 0;   // A/B switch
  bool f12_keep = getenv("SERL_AMD_CONV_F12_KEEP") == nullptr || strcmp(getenv("SERL_AMD_CONV_F12_KEEP"), "0") != 0;   // A/B switch
  bool actor_merge = getenv("SERL_AMD_ACTOR_MERGE") != nullptr && strcmp(getenv("SERL_AMD_ACTOR_MERGE"), "0") != 0;   // experiment (update loop)
  bool fuse_dgw1 = getenv("SERL_AMD_CONV_DGW1") == nullptr || strcmp(getenv("SERL_AMD_CONV_DGW1"), "0") != 0;   // A/B switch
  bool fuse12 = getenv("SERL_AMD_CONV_FUSE12") == nullptr || strcmp(getenv("SERL_AMD_CONV_FUSE12"), "0") != 0;   // A/B switch
  std::vector<TimeRec> trecs;
  std::vector<PhaseRec> precs;
};

namespace {

// ---------------------------------------------------------------- helpers mirroring serl_amd/ops.py
struct G {   // serl_gemm arguments with the defaults of ops.gemm
  const float* a; const float* b; float* c; const float* bias = nullptr;
  int nb = 1, M = 0, N = 0, K = 0;
  long long a_zs = 0, a_ms = 0, a_ks = 0, b_zs = 0, b_ks = 0, b_ns = 0, c_zs = 0, c_ms = 0, bias_zs = 0;
  int seglen = 0; long long a_ss = 0, b_ss = 0;
  int nsplit = 1; long long c_split = 0;
  int relu = 0, acc = 0, a_kc = 1, b_nc = 1;
};
int gemm(const G& g, hipStream_t st) {
  return serl_gemm(g.a, g.b, g.c, g.bias, g.nb, g.M, g.N, g.K, g.seglen ? g.seglen : g.K, g.a_zs, g.a_ss, g.a_ms, g.a_ks,
                   g.b_zs, g.b_ss, g.b_ks, g.b_ns, g.c_zs, g.c_ms, g.bias_zs, g.nsplit, g.c_split, g.relu, g.acc, g.a_kc,
                   g.b_nc, st);
}
int colsum(const float* x0, float* out0, int ng, int rpg, int N, int ldx, long long out_gs, hipStream_t st,
           const float* w = nullptr, const float* mul = nullptr, const float* x1 = nullptr, float* out1 = nullptr,
           const float* x2 = nullptr, const float* mul2 = nullptr, float* out2 = nullptr) {
  return serl_colsum(x0, w, mul, out0, x1, out1, x2, mul2, out2, ng, rpg, N, ldx, out_gs, st);
}

int ksplit(int K) {   // split-K factor for the skinny (M = batch) GEMMs so that >= ~256 blocks are in flight
  if (K < 1024) return 1;
  static const int nmax = getenv("SERL_AMD_KSPLIT_MAX") ? atoi(getenv("SERL_AMD_KSPLIT_MAX")) : 8;      // A/B switch
  static const int kmin = getenv("SERL_AMD_KSPLIT_MIN_K") ? atoi(getenv("SERL_AMD_KSPLIT_MIN_K")) : 256;
  int n = nmax;
  while (n > 1 && (K + n - 1) / n < kmin) n /= 2;
  const int klen = ((K + n - 1) / n + 15) / 16 * 16;
  return (K + klen - 1) / klen;
}

void bias_corrections(int step, float* bc1, float* bc2) {   // 1 - b^t in float32 like optax
  *bc1 = 1.0f - powf(ADAM_B1, (float)step);
  *bc2 = 1.0f - powf(ADAM_B2, (float)step);
}

void split2(const uint32_t k[2], uint32_t c0[2], uint32_t c1[2]) {   // c0, c1 = jax.random.split(k)
  uint32_t a0, a1, b0, b1;
  threefry_split_child(k[0], k[1], 2, 0, a0, a1);
  threefry_split_child(k[0], k[1], 2, 1, b0, b1);
  c0[0] = a0; c0[1] = a1; c1[0] = b0; c1[1] = b1;
}

bool split_conv_stack(const Cfg& c) {   // the conv stack runs on S-format tensors (include/serl_hip.h)
  if (serl_get_matmul_precision() != 1) return false;
  for (int li = 0; li < c.nl; ++li) {
    const ConvL& L = c.L[li];
    if (L.F % 8 || (li == 0 ? L.K * L.C > 32 : L.C % 8 != 0) || L.S > 2) return false;
  }
  return true;
}

void conv_cost(int kind, int in_u8, int n, int B, const ConvL& L, double* flops, double* bytes) {
  // algorithmic (flops, HBM bytes) of one grouped conv launch: every tensor read or written once
  const double x = (double)n * B * L.IH * L.IW * L.C * (in_u8 ? 1 : 4);
  const double y = (double)n * B * L.OH * L.OW * L.F * 4;
  const double w = (double)n * L.K * L.K * L.C * L.F * 4;
  *flops = 2.0 * n * B * L.OH * L.OW * L.F * L.K * L.K * L.C;
  if (kind == 5) {   // kind 3 with layer 1's output + byte masks kept for half of the groups (the critic pass next to the target pass)
    conv_cost(3, in_u8, n, B, L, flops, bytes);
    const int oh2 = (L.OH - 3) / 2 + 1, ow2 = (L.OW - 3) / 2 + 1;
    *bytes += 0.5 * n * ((double)B * L.OH * L.OW * L.F * (4 + 0.125) + (double)B * oh2 * ow2 * 64 * 0.125);
    return;
  }
  if (kind == 3) {   // layers 1 + 2 fused (L = layer 1, layer 2 = 3x3 stride 2 -> 64): frames in, layer-2 output out
    const int oh2 = (L.OH - 3) / 2 + 1, ow2 = (L.OW - 3) / 2 + 1, f2 = 64;
    *flops += 2.0 * n * B * oh2 * ow2 * f2 * 9.0 * L.F;
    *bytes = x + w + (double)n * 9 * L.F * f2 * 4 + (double)n * B * oh2 * ow2 * f2 * 4 + n * (L.F + f2) * 4.0;
    return;
  }
  if (kind == 4) {   // conv2's data gradient + conv1's weight gradient fused (L = layer 1): dz2, byte mask and frames in, dW1 out
    const int oh2 = (L.OH - 3) / 2 + 1, ow2 = (L.OW - 3) / 2 + 1, f2 = 64;
    *flops += 2.0 * n * B * oh2 * ow2 * f2 * 9.0 * L.F;
    *bytes = x + (double)n * B * oh2 * ow2 * f2 * 4 + (double)n * B * L.OH * L.OW * L.F / 8 + (double)n * 9 * L.F * f2 * 4 + w +
             n * L.F * 4.0;
    return;
  }
  if (kind == 0) *bytes = x + w + y + n * L.F * 4.0;
  else if (kind == 1) *bytes = y + w + x / 2 + x;
  else *bytes = x + y + w + n * L.F * 4.0;
}

struct Timed {   // RAII-free bracket: begin() before the launch, end() after, both on the launch stream
  serl_learner* h; hipStream_t st; int idx = -1;
  int begin(int kind, int in_u8, int n, int B, const ConvL& L) {
    if (!h->timing) return SERL_OK;
    TimeRec r{kind, in_u8, n, B, L.IH, L.C, L.F, 0, 0, nullptr, nullptr};
    conv_cost(kind, in_u8, n, B, L, &r.flops, &r.bytes);
    DRV_HIP(hipEventCreate(&r.a));
    DRV_HIP(hipEventCreate(&r.b));
    DRV_HIP(hipEventRecord(r.a, st));
    h->trecs.push_back(r);
    idx = (int)h->trecs.size() - 1;
    return SERL_OK;
  }
  int end() {
    if (idx >= 0) DRV_HIP(hipEventRecord(h->trecs[idx].b, st));
    return SERL_OK;
  }
};

// `to` waits for everything enqueued on `from` so far (no-op when they are the same stream)
int link(hipStream_t from, hipStream_t to, hipEvent_t ev) {
  if (from == to) return SERL_OK;
  DRV_HIP(ev_record(ev, from));
  DRV_HIP(ev_wait(to, ev));
  return SERL_OK;
}

int phase(serl_learner* h, const char* name, hipStream_t st) {
  if (!h->phases) return SERL_OK;
  PhaseRec r{name, nullptr};
  DRV_HIP(hipEventCreate(&r.ev));
  DRV_HIP(hipEventRecord(r.ev, st));
  h->precs.push_back(r);
  return SERL_OK;
}

// ================================================================= encoder (TwoD4PGEncoder + SpatialLearnedEmbeddings)
struct Pass { char tag; Store store; const EncOff* eo; const unsigned char* imgs[MAX_CAM]; };

size_t rmask_floats(int B, const ConvL& L) { return ((size_t)B * L.OH * L.OW * L.F / 8 + 3) / 4; }

int encoder_fwd(serl_learner* h, const Pass* passes, int npass, int B, Workspace& ws, hipStream_t st, char mask_tag,
                bool prep) {
  const Cfg& c = h->cfg;
  const int nl = c.nl, ncam = c.ncam;
  const bool split = split_conv_stack(c);
  struct Grp { const Pass* p; int ci; };
  std::vector<Grp> groups;
  for (int i = 0; i < npass; ++i)
    for (int ci = 0; ci < ncam; ++ci) groups.push_back(Grp{&passes[i], ci});
  const ConvL& last = c.L[nl - 1];
  for (size_t g0 = 0; g0 < groups.size(); g0 += MAXG_) {
    const int n = (int)std::min<size_t>(MAXG_, groups.size() - g0);
    const Grp* ch = &groups[g0];
    const void* xs[MAXG_];
    void* wf_all[MAX_CONV][MAXG_];
    for (int i = 0; i < n; ++i) xs[i] = ch[i].p->imgs[ch[i].ci];
    if (split) {
      const float* jw[32]; void* jwf[32]; void* jwd[32]; int ju8[32], jk[32], jc[32], jf[32];
      int nj = 0;
      auto flush = [&]() -> int {
        if (nj && prep) DRV(serl_convs_prep_weights_multi(nj, jw, jwf, jwd, ju8, jk, jk, jc, jf, st));
        nj = 0;
        return SERL_OK;
      };
      for (int li = 0; li < nl; ++li) {
        const ConvL& L = c.L[li];
        const long long nwf = serl_convs_weight_bytes(li == 0, L.K, L.K, L.C, L.F) / 4;
        for (int i = 0; i < n; ++i) {
          WSF(ws, st, wf, nwf, "wf%d_%c_%d", li, ch[i].p->tag, ch[i].ci);
          float* wd = nullptr;
          if (li > 0) {
            WSF(ws, st, wd_, (size_t)L.K * L.K * L.C * L.F, "wd%d_%c_%d", li, ch[i].p->tag, ch[i].ci);
            wd = wd_;
          }
          wf_all[li][i] = wf;
          jw[nj] = ch[i].p->store.at(ch[i].p->eo->ck[ch[i].ci][li]); jwf[nj] = wf; jwd[nj] = wd;
          ju8[nj] = li == 0; jk[nj] = L.K; jc[nj] = L.C; jf[nj] = L.F;
          if (++nj == 32) DRV(flush());
        }
      }
      DRV(flush());   // every (layer, group) in one launch, ahead of the dependent chain of conv kernels
    }
    // Passes that nobody differentiates (target critic, actor run-ahead, acting) do not need layer 1's output in
    // memory: their layers 1 + 2 run as one fused launch (serl_convs_fwd12), the pass whose masks are saved for the
    // backward (mask_tag) keeps the layer-by-layer launches.
    bool fused[MAXG_];
    int nfused = 0;
    {
      const bool fits = split && h->fuse12 && nl >= 2 && c.L[0].K == 3 && c.L[0].S == 2 && c.L[1].K == 3 && c.L[1].S == 2 &&
                        serl_convs_fwd12_supported(c.L[0].IH, c.L[0].IW, c.L[0].C, c.L[0].F, c.L[1].F);
      // ... unless the fused kernel can keep layer 1's output for it (serl_convs_fwd12_keep): then every pass is fused
      const bool keep = fits && h->f12_keep && serl_convs_fwd12_keep_supported(c.L[0].IH, c.L[0].IW, c.L[0].C, c.L[0].F, c.L[1].F);
      for (int i = 0; i < n; ++i) {
        fused[i] = fits && (keep || !(mask_tag && ch[i].p->tag == mask_tag));
        nfused += fused[i];
      }
    }
    for (int li = 0; li < nl; ++li) {
      const ConvL& L = c.L[li];
      // groups of this layer's plain launch (all of them from layer 2 on)
      int idx[MAXG_], m = 0;
      for (int i = 0; i < n; ++i)
        if (li >= 2 || !fused[i]) idx[m++] = i;
      const void* xin[MAXG_]; const void* wfp[MAXG_];
      const float* wsp[MAXG_]; const float* bsp[MAXG_]; void* ys[MAXG_]; void* rm[MAXG_];
      bool any_mask = false;
      for (int k = 0; k < m; ++k) {
        const int i = idx[k];
        const Pass* p = ch[i].p;
        xin[k] = xs[i];
        wfp[k] = split ? wf_all[li][i] : nullptr;
        wsp[k] = p->store.at(p->eo->ck[ch[i].ci][li]);
        bsp[k] = p->store.at(p->eo->cb[ch[i].ci][li]);
        WSF(ws, st, y, (size_t)B * L.OH * L.OW * L.F, "act%d_%c_%d", li, p->tag, ch[i].ci);
        ys[k] = y;
        rm[k] = nullptr;
        if (split && mask_tag && li < nl - 1 && p->tag == mask_tag) {
          WSF(ws, st, r, rmask_floats(B, L), "rmask%d_%c_%d", li, p->tag, ch[i].ci);
          rm[k] = r;
          any_mask = true;
        }
      }
      if (m > 0) {
        Timed t{h, st};
        DRV(t.begin(0, li == 0, m, B, L));
        if (split)
          DRV(serl_convs_fwd(m, xin, wfp, bsp, ys, li == 0, B, L.IH, L.IW, L.C, L.K, L.K, L.S, L.F, 1,
                             any_mask ? rm : nullptr, st));
        else
          DRV(serl_conv2d_fwd(m, xin, wsp, bsp, (float* const*)ys, li == 0, B, L.IH, L.IW, L.C, L.K, L.K, L.S, L.F, 1, st));
        DRV(t.end());
        for (int k = 0; k < m; ++k) xs[idx[k]] = ys[k];
      }
      if (li == 0 && nfused > 0) {
        const ConvL& L1 = c.L[1];
        const void* fx[MAXG_]; const void* f1[MAXG_]; const void* f2[MAXG_]; const float* fb1[MAXG_]; const float* fb2[MAXG_];
        void* fy[MAXG_]; void* fy1[MAXG_]; void* frm1[MAXG_]; void* frm2[MAXG_];
        int fi[MAXG_], k = 0, nkeep = 0;
        for (int i = 0; i < n; ++i) {
          if (!fused[i]) continue;
          const Pass* p = ch[i].p;
          fx[k] = xs[i]; f1[k] = wf_all[0][i]; f2[k] = wf_all[1][i];
          fb1[k] = p->store.at(p->eo->cb[ch[i].ci][0]); fb2[k] = p->store.at(p->eo->cb[ch[i].ci][1]);
          WSF(ws, st, y, (size_t)B * L1.OH * L1.OW * L1.F, "act1_%c_%d", p->tag, ch[i].ci);
          fy[k] = y; fi[k] = i;
          fy1[k] = nullptr; frm1[k] = nullptr; frm2[k] = nullptr;
          if (mask_tag && p->tag == mask_tag) {      // the pass with a backward: layer 1's output and both ReLU masks
            WSF(ws, st, y0, (size_t)B * L.OH * L.OW * L.F, "act0_%c_%d", p->tag, ch[i].ci);
            WSF(ws, st, r0, rmask_floats(B, L), "rmask0_%c_%d", p->tag, ch[i].ci);
            fy1[k] = y0; frm1[k] = r0;
            if (nl > 2) {
              WSF(ws, st, r1, rmask_floats(B, L1), "rmask1_%c_%d", p->tag, ch[i].ci);
              frm2[k] = r1;
            }
            ++nkeep;
          }
          ++k;
        }
        Timed t{h, st};
        DRV(t.begin(nkeep ? 5 : 3, 1, k, B, L));
        if (nkeep) DRV(serl_convs_fwd12_keep(k, fx, f1, fb1, f2, fb2, fy, B, L.IH, L.IW, L.C, L.F, L1.F, 1, frm2, fy1, frm1, st));
        else DRV(serl_convs_fwd12(k, fx, f1, fb1, f2, fb2, fy, B, L.IH, L.IW, L.C, L.F, L1.F, 1, nullptr, st));
        DRV(t.end());
        for (int q = 0; q < k; ++q) xs[fi[q]] = fy[q];
      }
    }
    const void* feats[MAXG_]; const float* kers[MAXG_]; float* outs[MAXG_];
    for (int i = 0; i < n; ++i) {
      const Pass* p = ch[i].p;
      feats[i] = xs[i];
      kers[i] = p->store.at(p->eo->sle[ch[i].ci]);
      WSF(ws, st, enc, (size_t)B * c.enc_dim, "enc_%c", p->tag);
      outs[i] = enc + (size_t)ch[i].ci * last.F * SLE_FEATURES;
    }
    DRV(serl_sle_fwd(n, feats, kers, outs, c.enc_dim, B, last.OH * last.OW, last.F, SLE_FEATURES, split, st));
  }
  return SERL_OK;
}

int encoder_bwd(serl_learner* h, char tag, const Store& store, const EncOff& eo, const Store& grads,
                const unsigned char* const* imgs, int B, Workspace& ws, hipStream_t st, GradSide* gside, bool use_masks) {
  const Cfg& c = h->cfg;
  const int nl = c.nl, ncam = c.ncam;
  const ConvL& last = c.L[nl - 1];
  const bool split = split_conv_stack(c);
  WSF(ws, st, denc, (size_t)B * c.enc_dim, "denc_%c", tag);
  const float* douts[MAX_CAM]; const float* kers[MAX_CAM]; const void* feats[MAX_CAM]; void* dfeat[MAX_CAM]; float* dker[MAX_CAM];
  for (int ci = 0; ci < ncam; ++ci) {
    douts[ci] = denc + (size_t)ci * last.F * SLE_FEATURES;
    kers[ci] = store.at(eo.sle[ci]);
    WSF(ws, st, f, (size_t)B * last.OH * last.OW * last.F, "act%d_%c_%d", nl - 1, tag, ci);
    feats[ci] = f;
    WSF(ws, st, df, (size_t)B * last.OH * last.OW * last.F, "dact%d_%d", nl - 1, ci);
    dfeat[ci] = df;
    dker[ci] = grads.at(eo.sle[ci]);
  }
  const int HW = last.OH * last.OW;
  if (!gside) {
    DRV(serl_sle_bwd(ncam, douts, c.enc_dim, kers, feats, dfeat, dker, B, HW, last.F, SLE_FEATURES, split, st));
  } else {
    hipStream_t sw;
    GradSide* const gs_sle = gside == &h->gside && h->wg_streams == 3 ? &h->gside2 : gside;   // (with the heads' gradients)
    DRV(gs_sle->fork(&sw));
    DRV(serl_sle_bwd(ncam, douts, c.enc_dim, kers, feats, nullptr, dker, B, HW, last.F, SLE_FEATURES, split, sw));
    DRV(serl_sle_bwd(ncam, douts, c.enc_dim, kers, feats, dfeat, nullptr, B, HW, last.F, SLE_FEATURES, split, st));
  }
  long long slab_bytes = 0;
  for (int li = 0; li < nl; ++li) {
    const ConvL& L = c.L[li];
    const long long b = serl_conv2d_wgrad_workspace_bytes(ncam, B, L.IH, L.IW, L.C, L.K, L.K, L.S, L.F);
    SERL_CHECK_ARG(b >= 0, "learner: bad conv geometry in layer %d", li);
    slab_bytes = std::max(slab_bytes, b);
  }
  // conv2's data gradient and conv1's weight gradient in one launch (serl_convs_dgrad_wgrad1): dz1 stays on chip
  const bool dgw1_fits = split && nl >= 2 && c.L[0].K == 3 && c.L[0].S == 2 && c.L[1].K == 3 && c.L[1].S == 2 &&
                         serl_convs_dgrad_wgrad1_supported(c.L[0].IH, c.L[0].IW, c.L[0].C, c.L[0].F, c.L[1].F);
  const bool dgw1 = dgw1_fits && use_masks && h->fuse_dgw1;
  if (dgw1_fits) slab_bytes = std::max(slab_bytes, serl_convs_dgrad_wgrad1_workspace_bytes(ncam));   // (one size per name)
  WSF(ws, st, slab, (size_t)slab_bytes / 4, "wgrad_slab");
  for (int li = nl - 1; li >= 0; --li) {
    const ConvL& L = c.L[li];
    if (li == 0 && dgw1) break;     // done by the fused launch of layer 2's data gradient
    const void* dz[MAX_CAM]; const void* xs[MAX_CAM]; float* dw[MAX_CAM]; float* db[MAX_CAM];
    for (int ci = 0; ci < ncam; ++ci) {
      WSF(ws, st, d, (size_t)B * L.OH * L.OW * L.F, "dact%d_%d", li, ci);
      dz[ci] = d;
      if (li == 0) {
        xs[ci] = imgs[ci];
      } else {
        const ConvL& P = c.L[li - 1];
        WSF(ws, st, x, (size_t)B * P.OH * P.OW * P.F, "act%d_%c_%d", li - 1, tag, ci);
        xs[ci] = x;
      }
      dw[ci] = grads.at(eo.ck[ci][li]);
      db[ci] = grads.at(eo.cb[ci][li]);
    }
    // The side stream carries more work than the main stream's data-gradient chain: the first layer's weight gradient,
    // which becomes ready last, runs on the main stream behind the last data gradient (its own slabs).
    const bool on_main = gside && li < h->wg_main_layers && nl > 1;
    hipStream_t sw = st;
    float* slab_l = slab;
    if (gside && !on_main) {                        // dz of this layer is ready on the main stream here
      if (gside == &h->gside && h->wg_streams == 2 && ((nl - 1 - li) & 1)) {
        DRV(h->gside2.fork(&sw));
        WSF(ws, st, slab2, (size_t)slab_bytes / 4, "wgrad_slab2");
        slab_l = slab2;
      } else {
        DRV(gside->fork(&sw));
      }
    }
    if (on_main) {
      WSF(ws, st, slab1, (size_t)slab_bytes / 4, "wgrad_slab_main");
      slab_l = slab1;
    }
    {
      Timed t{h, sw};
      DRV(t.begin(2, li == 0, ncam, B, L));
      if (split)
        DRV(serl_convs_wgrad(ncam, xs, dz, dw, db, slab_l, slab_bytes, li == 0, B, L.IH, L.IW, L.C, L.K, L.K, L.S, L.F, sw));
      else
        DRV(serl_conv2d_wgrad(ncam, xs, (const float* const*)dz, dw, db, slab_l, slab_bytes, li == 0, B, L.IH, L.IW, L.C,
                              L.K, L.K, L.S, L.F, sw));
      DRV(t.end());
    }
    if (li == 1 && dgw1) {
      const ConvL& P = c.L[0];
      const void* wds[MAX_CAM]; const void* rm[MAX_CAM]; const void* fr[MAX_CAM]; float* dw0[MAX_CAM]; float* db0[MAX_CAM];
      for (int ci = 0; ci < ncam; ++ci) {
        WSF(ws, st, wd, (size_t)L.K * L.K * L.C * L.F, "wd%d_%c_%d", li, tag, ci);
        wds[ci] = wd;
        WSF(ws, st, r, rmask_floats(B, P), "rmask%d_%c_%d", 0, tag, ci);
        rm[ci] = r;
        fr[ci] = imgs[ci];
        dw0[ci] = grads.at(eo.ck[ci][0]);
        db0[ci] = grads.at(eo.cb[ci][0]);
      }
      float* slab_f = slab;
      if (gside) {                    // the side streams' weight gradients are using `slab`
        WSF(ws, st, slab1, (size_t)slab_bytes / 4, "wgrad_slab_main");
        slab_f = slab1;
      }
      Timed t{h, st};
      DRV(t.begin(4, 1, ncam, B, P));
      DRV(serl_convs_dgrad_wgrad1(ncam, dz, wds, rm, fr, dw0, db0, slab_f, slab_bytes, B, P.IH, P.IW, P.C, P.F, L.F, st));
      DRV(t.end());
    } else if (li > 0) {
      const ConvL& P = c.L[li - 1];
      void* dx[MAX_CAM];
      for (int ci = 0; ci < ncam; ++ci) {
        WSF(ws, st, d, (size_t)B * P.OH * P.OW * P.F, "dact%d_%d", li - 1, ci);
        dx[ci] = d;
      }
      Timed t{h, st};
      DRV(t.begin(1, 0, ncam, B, L));
      if (split) {
        const void* wds[MAX_CAM]; const void* rm[MAX_CAM];
        for (int ci = 0; ci < ncam; ++ci) {
          WSF(ws, st, wd, (size_t)L.K * L.K * L.C * L.F, "wd%d_%c_%d", li, tag, ci);
          wds[ci] = wd;
          if (use_masks) {
            WSF(ws, st, r, rmask_floats(B, P), "rmask%d_%c_%d", li - 1, tag, ci);
            rm[ci] = r;
          }
        }
        DRV(serl_convs_dgrad(ncam, dz, wds, use_masks ? rm : xs, dx, B, L.IH, L.IW, L.C, L.K, L.K, L.S, L.F,
                             use_masks ? 1 : 0, st));
      } else {
        const float* wk[MAX_CAM];
        for (int ci = 0; ci < ncam; ++ci) wk[ci] = store.at(eo.ck[ci][li]);
        DRV(serl_conv2d_dgrad(ncam, (const float* const*)dz, wk, (const float* const*)xs, (float* const*)dx, B, L.IH,
                              L.IW, L.C, L.K, L.K, L.S, L.F, st));
      }
      DRV(t.end());
    }
  }
  return SERL_OK;
}

// ================================================================= trunk (PixelMultiplexer Dense/LN/tanh branches)
int trunk_fwd(serl_learner* h, char tag, const Store& s, const EncOff& eo, const float* state_ptr, float* z, int ldz, int B,
              Workspace& ws, hipStream_t st, bool save) {
  const Cfg& c = h->cfg;
  const int K = c.enc_dim, ns = ksplit(K), Ld = c.latent_dim, Sd = c.state_dim;
  WSF(ws, st, slab, (size_t)ns * B * TRUNK_DIM, "slab0_%c", tag);
  WSF(ws, st, enc, (size_t)B * K, "enc_%c", tag);
  G g; g.a = enc; g.b = s.at(eo.d0k); g.c = slab; g.M = B; g.N = TRUNK_DIM; g.K = K; g.a_ms = K; g.a_ks = 1;
  g.b_ks = TRUNK_DIM; g.b_ns = 1; g.c_ms = TRUNK_DIM; g.nsplit = ns; g.c_split = (long long)B * TRUNK_DIM;
  DRV(gemm(g, st));
  float *xh0 = nullptr, *rs0 = nullptr, *xh1 = nullptr, *rs1 = nullptr;
  if (save) {
    WSF(ws, st, a, (size_t)B * TRUNK_DIM, "xhat0_%c", tag); xh0 = a;
    WSF(ws, st, b, (size_t)B, "rstd0_%c", tag); rs0 = b;
    WSF(ws, st, a1, (size_t)B * Ld, "xhat1_%c", tag); xh1 = a1;
    WSF(ws, st, b1, (size_t)B, "rstd1_%c", tag); rs1 = b1;
  }
  return serl_trunk_ln_fwd(slab, ns, (long long)B * TRUNK_DIM, TRUNK_DIM, s.at(eo.d0b), s.at(eo.l0s), s.at(eo.l0b), z, xh0,
                           rs0, state_ptr, Sd, Sd, s.at(eo.d1k), Ld, s.at(eo.d1b), s.at(eo.l1s), s.at(eo.l1b),
                           z + TRUNK_DIM, xh1, rs1, ldz, B, ACT_TANH, st);
}

int trunk_bwd(serl_learner* h, char tag, const Store& s, const EncOff& eo, const Store& gr, const float* state_ptr,
              const float* z, int ldz, const float* dz, int lddz, int B, Workspace& ws, hipStream_t st, bool need_denc,
              GradSide* gside, int dz_nslab, long long dz_ss) {
  const Cfg& c = h->cfg;
  const int K = c.enc_dim, Ld = c.latent_dim, Sd = c.state_dim;
  WSF(ws, st, dpre0, (size_t)B * TRUNK_DIM, "dpre0");
  WSF(ws, st, g0, (size_t)B * TRUNK_DIM, "g0");
  WSF(ws, st, xh0, (size_t)B * TRUNK_DIM, "xhat0_%c", tag);
  WSF(ws, st, rs0, (size_t)B, "rstd0_%c", tag);
  WSF(ws, st, dpre1, (size_t)B * Ld, "dpre1");
  WSF(ws, st, g1, (size_t)B * Ld, "g1");
  WSF(ws, st, xh1, (size_t)B * Ld, "xhat1_%c", tag);
  WSF(ws, st, rs1, (size_t)B, "rstd1_%c", tag);
  const float* dy[2] = {dz, dz + TRUNK_DIM}; int lddy[2] = {lddz, lddz};
  const float* y[2] = {z, z + TRUNK_DIM}; int ldy[2] = {ldz, ldz};
  const float* xh[2] = {xh0, xh1}; const float* rs[2] = {rs0, rs1};
  const float* gam[2] = {s.at(eo.l0s), s.at(eo.l1s)};
  float* dzo[2] = {dpre0, dpre1}; float* go[2] = {g0, g1}; int N[2] = {TRUNK_DIM, Ld};
  DRV(serl_ln_act_bwd_seg(2, dy, lddy, dz_nslab, dz_ss, y, ldy, xh, rs, gam, dzo, go, B, N, ACT_TANH, st));
  hipStream_t stw = st;
  if (gside) DRV(gside->fork(&stw));
  DRV(colsum(g0, gr.at(eo.l0b), 1, B, TRUNK_DIM, TRUNK_DIM, 0, stw, nullptr, nullptr, dpre0, gr.at(eo.d0b), g0, xh0,
             gr.at(eo.l0s)));
  WSF(ws, st, enc, (size_t)B * K, "enc_%c", tag);
  {
    G g; g.a = enc; g.b = dpre0; g.c = gr.at(eo.d0k); g.M = K; g.N = TRUNK_DIM; g.K = B; g.a_ms = 1; g.a_ks = K;
    g.b_ks = TRUNK_DIM; g.b_ns = 1; g.c_ms = TRUNK_DIM; g.a_kc = 0;
    DRV(gemm(g, stw));
  }
  if (need_denc) {
    WSF(ws, st, denc, (size_t)B * K, "denc_%c", tag);
    G g; g.a = dpre0; g.b = s.at(eo.d0k); g.c = denc; g.M = B; g.N = K; g.K = TRUNK_DIM; g.a_ms = TRUNK_DIM; g.a_ks = 1;
    g.b_ks = 1; g.b_ns = TRUNK_DIM; g.c_ms = K; g.b_nc = 0;
    DRV(gemm(g, st));
  }
  DRV(colsum(g1, gr.at(eo.l1b), 1, B, Ld, Ld, 0, stw, nullptr, nullptr, dpre1, gr.at(eo.d1b), g1, xh1, gr.at(eo.l1s)));
  G g; g.a = state_ptr; g.b = dpre1; g.c = gr.at(eo.d1k); g.M = Sd; g.N = Ld; g.K = B; g.a_ms = 1; g.a_ks = Sd;
  g.b_ks = Ld; g.b_ns = 1; g.c_ms = Ld; g.a_kc = 0;
  return gemm(g, stw);
}

// ================================================================= critic ensemble (Ensemble(StateActionValue(MLP)))
// members m0, m0 + mstride, ... (E of them) on a shared input (B, ld_inp); *q_out = (E, B)
// nn.Dropout behind hidden layer `layer` of the critic MLPs (mlp.py:26-29) on (E, B, hd) pre-activations or their
// gradients: member position j takes bernoulli(fold_in(fold_in(key, layer), j), keep) over the (global rows, hd) array
// (BUILD-DEFINED stream, include/serl_hip.h serl_learner_config.critic_dropout); the same call masks the gradient.
int critic_dropout(const Cfg& c, const uint32_t* key, int layer, float* x, int E, int B, int hd, int row0, int total,
                   hipStream_t st) {
  unsigned kl[2], km[2];
  DRV(serl_prng_fold_in(key[0], key[1], (unsigned)layer, kl));
  for (int j = 0; j < E; ++j) {
    DRV(serl_prng_fold_in(kl[0], kl[1], (unsigned)j, km));
    DRV(serl_dropout(x + (size_t)j * B * hd, hd, B, hd, km[0], km[1], 1.0f - c.crit_drop, row0, total, st));
  }
  return SERL_OK;
}

bool ens_fwd_fusable(const serl_learner* h) {
  const Cfg& c = h->cfg;
  return h->ens_fused && c.nh == 2 && c.hid[0] == 256 && c.hid[1] == 256 && serl_get_matmul_precision() == 1;
}
// arguments (and workspace buffers) of one fused ensemble forward: members m0, m0 + mstride, ... (E of them)
int ens_fwd_job(serl_learner* h, char tag, const Store& s, const CriticOff& co, int m0, int mstride, int E, const float* inp,
                int ld_inp, int B, Workspace& ws, hipStream_t st, bool save, serl_ens_fwd_job* j) {
  const Cfg& c = h->cfg;
  const int din = c.latent + c.crit_A;
  const long long v = 256;
  WSF(ws, st, q, (size_t)E * B, "q_%c", tag);
  float *h0 = nullptr, *h1 = nullptr, *xh0 = nullptr, *xh1 = nullptr, *rs0 = nullptr, *rs1 = nullptr;
  if (save) {
    WSF(ws, st, a0, (size_t)E * B * 256, "h0_%c", tag); h0 = a0;
    WSF(ws, st, a1, (size_t)E * B * 256, "h1_%c", tag); h1 = a1;
    if (c.ln) {
      WSF(ws, st, x0, (size_t)E * B * 256, "exh0_%c", tag); xh0 = x0;
      WSF(ws, st, x1, (size_t)E * B * 256, "exh1_%c", tag); xh1 = x1;
      WSF(ws, st, r0, (size_t)E * B, "ers0_%c", tag); rs0 = r0;
      WSF(ws, st, r1, (size_t)E * B, "ers1_%c", tag); rs1 = r1;
    }
  }
  *j = serl_ens_fwd_job{inp, ld_inp, din, B, E,
                        s.at(co.w[0]) + (long long)m0 * din * 256, s.at(co.b[0]) + m0 * v,
                        c.ln ? s.at(co.lns[0]) + m0 * v : nullptr, c.ln ? s.at(co.lnb[0]) + m0 * v : nullptr,
                        s.at(co.w[1]) + (long long)m0 * 256 * 256, s.at(co.b[1]) + m0 * v,
                        c.ln ? s.at(co.lns[1]) + m0 * v : nullptr, c.ln ? s.at(co.lnb[1]) + m0 * v : nullptr,
                        s.at(co.hk) + (long long)m0 * 256, s.at(co.hb) + m0,
                        (long long)mstride * din * 256, (long long)mstride * 256 * 256, mstride * v, (long long)mstride,
                        c.ln, h0, h1, xh0, xh1, rs0, rs1, q};
  return SERL_OK;
}

// drop: rngs={"dropout": key} of a training=True apply (used only with critic_dropout_rate > 0); row0 / total: global row
// of this rank's row 0 and the global minibatch size
int ensemble_fwd(serl_learner* h, char tag, const Store& s, const CriticOff& co, int m0, int mstride, int E,
                 const float* inp, int ld_inp, int B, Workspace& ws, hipStream_t st, bool save, float** q_out,
                 const uint32_t* drop = nullptr, int row0 = 0, int total = 0) {
  const Cfg& c = h->cfg;
  if (c.crit_drop <= 0.f) drop = nullptr;
  int din = c.latent + c.crit_A;
  const int nl = c.nh, hl = c.hid[nl - 1];
  const float* a_ptr = inp; long long a_zs = 0, a_ms = ld_inp;
  WSF(ws, st, q, (size_t)E * B, "q_%c", tag);
  const float* head_w = s.at(co.hk) + (long long)m0 * hl;
  const float* head_b = s.at(co.hb) + m0;
  if (!drop && ens_fwd_fusable(h)) {
    // the whole pass in one launch (ensemble.hip): four dependent launches less on a latency-bound chain
    serl_ens_fwd_job j;
    DRV(ens_fwd_job(h, tag, s, co, m0, mstride, E, inp, ld_inp, B, ws, st, save, &j));
    DRV(serl_ensemble_fwd_fused(j.inp, j.ld_inp, j.din, j.B, j.E, j.w0, j.b0, j.g0, j.be0, j.w1, j.b1, j.g1, j.be1, j.hw, j.hb,
                                j.w0_member_stride, j.w1_member_stride, j.vec_member_stride, j.head_bias_member_stride,
                                j.use_ln, j.h0, j.h1, j.xh0, j.xh1, j.rs0, j.rs1, j.q, st));
    *q_out = j.q;
    return SERL_OK;
  }
  for (int i = 0; i < nl; ++i) {
    const int hd = c.hid[i];
    const float* W = s.at(co.w[i]) + (long long)m0 * din * hd;
    const float* bias = s.at(co.b[i]) + (long long)m0 * hd;
    WSF(ws, st, hbuf, (size_t)E * B * hd, "h%d_%c", i, tag);
    if (c.ln) {
      WSF(ws, st, pbuf, (size_t)E * B * hd, "epre%d_%c", i, tag);
      G g; g.a = a_ptr; g.b = W; g.c = pbuf; g.nb = E; g.M = B; g.N = hd; g.K = din; g.a_zs = a_zs; g.a_ms = a_ms; g.a_ks = 1;
      g.b_zs = (long long)mstride * din * hd; g.b_ks = hd; g.b_ns = 1; g.c_zs = (long long)B * hd; g.c_ms = hd;
      if (drop) { g.bias = bias; g.bias_zs = (long long)mstride * hd; }   // Dense (with bias) -> Dropout -> LayerNorm
      DRV(gemm(g, st));
      if (drop) {
        DRV(critic_dropout(c, drop, i, pbuf, E, B, hd, row0, total, st));
        bias = nullptr;
      }
      float *xh = nullptr, *rs = nullptr;
      if (save) {
        WSF(ws, st, a, (size_t)E * B * hd, "exh%d_%c", i, tag); xh = a;
        WSF(ws, st, b, (size_t)E * B, "ers%d_%c", i, tag); rs = b;
      }
      const float* gam = s.at(co.lns[i]) + (long long)m0 * hd;
      const float* bet = s.at(co.lnb[i]) + (long long)m0 * hd;
      if (i == nl - 1)   // the last hidden layer's launch also applies the Dense(1) Q head
        DRV(serl_ln_act_head_fwd(pbuf, 1, 0, hd, bias, gam, bet, (long long)mstride * hd, hbuf, hd, xh, rs, E * B, hd, B, 1,
                                 ACT_RELU, head_w, (long long)mstride * hl, head_b, mstride, q, st));
      else
        DRV(serl_ln_act_fwd(pbuf, 1, 0, hd, bias, gam, bet, (long long)mstride * hd, hbuf, hd, xh, rs, E * B, hd, B, 1,
                            ACT_RELU, st));
    } else {
      G g; g.a = a_ptr; g.b = W; g.c = hbuf; g.bias = bias; g.nb = E; g.M = B; g.N = hd; g.K = din; g.a_zs = a_zs;
      g.a_ms = a_ms; g.a_ks = 1; g.b_zs = (long long)mstride * din * hd; g.b_ks = hd; g.b_ns = 1; g.c_zs = (long long)B * hd;
      g.c_ms = hd; g.bias_zs = (long long)mstride * hd; g.relu = drop ? 0 : 1;
      DRV(gemm(g, st));
      if (drop) {                                                          // Dense -> Dropout -> ReLU
        DRV(critic_dropout(c, drop, i, hbuf, E, B, hd, row0, total, st));
        DRV(serl_ln_act_fwd(hbuf, 1, 0, hd, nullptr, nullptr, nullptr, 0, hbuf, hd, nullptr, nullptr, E * B, hd, B, 0, ACT_RELU, st));
      }
    }
    a_ptr = hbuf; a_zs = (long long)B * hd; a_ms = hd; din = hd;
  }
  if (!c.ln) DRV(serl_rowdot(a_ptr, head_w, (long long)mstride * hl, head_b, mstride, q, E * B, din, B, st));
  *q_out = q;
  return SERL_OK;
}

// Backward of the full ensemble from dq (E, B).  Parameter gradients into `grads` when given.  want_dinp: 0 none,
// 1 summed (B, latent+action), 2 split-K slabs (the consumer sums them on read).  Result in dinp / nslab / slab_stride.
int ensemble_bwd(serl_learner* h, char tag, const Store& s, const CriticOff& co, const Store* grads, const float* dq,
                 const float* inp, int ld_inp, int B, Workspace& ws, hipStream_t st, GradSide* gside, int want_dinp,
                 float** dinp_out, int* nslab_out, long long* ss_out, const uint32_t* drop = nullptr, int row0 = 0,
                 int total = 0) {
  const Cfg& c = h->cfg;
  if (c.crit_drop <= 0.f) drop = nullptr;
  const int E = c.E, ln = c.ln, nl = c.nh;
  int dims[MAX_HID + 1];
  dims[0] = c.latent + c.crit_A;
  for (int i = 0; i < nl; ++i) dims[i + 1] = c.hid[i];
  const int hl = dims[nl];
  hipStream_t stw = st;
  const bool fused_ = !drop && h->ens_bwd_fused && nl == 2 && c.hid[0] == 256 && c.hid[1] == 256 && serl_get_matmul_precision() == 1;
  if (grads && !fused_) {
    if (gside) DRV(gside->fork(&stw));
    WSF(ws, st, hlast, (size_t)E * B * hl, "h%d_%c", nl - 1, tag);
    DRV(colsum(hlast, grads->at(co.hk), E, B, hl, hl, hl, stw, dq));
    DRV(colsum(dq, grads->at(co.hb), E, B, 1, 1, 1, stw));
  }
  if (fused_) {
    // one launch for ln_act_bwd + GEMM + ln_act_bwd, one for every batch-column sum (csrc/ensemble.hip)
    const size_t nel = (size_t)E * B * 256;
    WSF(ws, st, h0, nel, "h0_%c", tag);
    WSF(ws, st, h1, nel, "h1_%c", tag);
    WSF(ws, st, dz1, nel, "edz1");
    WSF(ws, st, dz0, nel, "edz0");
    float *g1 = nullptr, *g0 = nullptr, *xh0 = nullptr, *xh1 = nullptr, *rs0 = nullptr, *rs1 = nullptr;
    if (ln) {
      WSF(ws, st, a1, nel, "eg1"); g1 = a1;
      WSF(ws, st, a0, nel, "eg0"); g0 = a0;
      WSF(ws, st, x0, nel, "exh0_%c", tag); xh0 = x0;
      WSF(ws, st, x1, nel, "exh1_%c", tag); xh1 = x1;
      WSF(ws, st, r0, (size_t)E * B, "ers0_%c", tag); rs0 = r0;
      WSF(ws, st, r1, (size_t)E * B, "ers1_%c", tag); rs1 = r1;
    }
    DRV(serl_ensemble_bwd_fused(dq, B, E, s.at(co.hk), s.at(co.w[1]), ln ? s.at(co.lns[0]) : nullptr,
                                ln ? s.at(co.lns[1]) : nullptr, ln, h0, h1, xh0, xh1, rs0, rs1, dz1, g1, dz0, g0, st));
    if (grads) {
      if (gside) DRV(gside->fork(&stw));
      const bool w0_here = dims[0] <= 32;   // state-based learners: the first kernel's gradient rides in the column-sum launch
      DRV(serl_ensemble_colsum(dq, h1, g1, xh1, dz1, g0, xh0, dz0, B, E, ln, grads->at(co.hk), grads->at(co.hb),
                               ln ? grads->at(co.lnb[1]) : nullptr, ln ? grads->at(co.lns[1]) : nullptr, grads->at(co.b[1]),
                               ln ? grads->at(co.lnb[0]) : nullptr, ln ? grads->at(co.lns[0]) : nullptr, grads->at(co.b[0]),
                               w0_here ? inp : nullptr, ld_inp, dims[0], w0_here ? grads->at(co.w[0]) : nullptr, stw));
      for (int i = 1; i >= (w0_here ? 1 : 0); --i) {      // dW_i[e] = (input of layer i)^T . dz_i[e]
        const int hp = dims[i];
        const float* a_ptr = i == 0 ? inp : h0;
        G gg; gg.a = a_ptr; gg.b = i == 0 ? dz0 : dz1; gg.c = grads->at(co.w[i]); gg.nb = E; gg.M = hp; gg.N = 256; gg.K = B;
        gg.a_zs = i == 0 ? 0 : (long long)B * 256; gg.a_ms = 1; gg.a_ks = i == 0 ? ld_inp : 256; gg.b_zs = (long long)B * 256;
        gg.b_ks = 256; gg.b_ns = 1; gg.c_zs = (long long)hp * 256; gg.c_ms = 256; gg.a_kc = 0;
        DRV(gemm(gg, stw));
      }
    }
    if (want_dinp == 0) return SERL_OK;
    const int hp = dims[0], hd = 256;
    const float* W = s.at(co.w[0]);
    const float* dz = dz0;
    WSF(ws, st, dinp, (size_t)B * hp, "dinp");
    const int ns = ksplit(E * hd);
    G gg; gg.a = dz; gg.b = W; gg.M = B; gg.N = hp; gg.K = E * hd; gg.seglen = hd; gg.a_ss = (long long)B * hd; gg.a_ms = hd;
    gg.a_ks = 1; gg.b_ss = (long long)hp * hd; gg.b_ks = 1; gg.b_ns = hd; gg.c_ms = hp; gg.b_nc = 0;
    if (ns > 1) {
      WSF(ws, st, slab, (size_t)ns * B * hp, "dinp_slab");
      gg.c = slab; gg.nsplit = ns; gg.c_split = (long long)B * hp;
      DRV(gemm(gg, st));
      if (want_dinp == 2) { *dinp_out = slab; *nslab_out = ns; *ss_out = (long long)B * hp; return SERL_OK; }
      DRV(serl_ln_act_fwd(slab, ns, (long long)B * hp, hp, nullptr, nullptr, nullptr, 0, dinp, hp, nullptr, nullptr, B, hp,
                          B, 0, ACT_NONE, st));
    } else {
      gg.c = dinp;
      DRV(gemm(gg, st));
    }
    *dinp_out = dinp; *nslab_out = 1; *ss_out = 0;
    return SERL_OK;
  }
  const float* dy = nullptr;
  for (int i = nl - 1; i >= 0; --i) {
    const int hd = dims[i + 1], hp = dims[i];
    WSF(ws, st, hbuf, (size_t)E * B * hd, "h%d_%c", i, tag);
    WSF(ws, st, dz, (size_t)E * B * hd, "edz%d", i);
    float *g = nullptr, *xh = nullptr, *rs = nullptr; const float* gam = nullptr;
    if (ln) {
      WSF(ws, st, g_, (size_t)E * B * hd, "eg%d", i); g = g_;
      WSF(ws, st, x_, (size_t)E * B * hd, "exh%d_%c", i, tag); xh = x_;
      WSF(ws, st, r_, (size_t)E * B, "ers%d_%c", i, tag); rs = r_;
      gam = s.at(co.lns[i]);
    }
    if (i == nl - 1)
      DRV(serl_ln_act_bwd(nullptr, 0, dq, s.at(co.hk), hd, hbuf, hd, xh, rs, gam, hd, dz, g, E * B, hd, B, ln, ACT_RELU, st));
    else
      DRV(serl_ln_act_bwd(dy, hd, nullptr, nullptr, 0, hbuf, hd, xh, rs, gam, hd, dz, g, E * B, hd, B, ln, ACT_RELU, st));
    if (drop) DRV(critic_dropout(c, drop, i, dz, E, B, hd, row0, total, st));   // gradient w.r.t. the Dense output
    const float* W = s.at(co.w[i]);
    if (grads) {
      float* gb = grads->at(co.b[i]);
      if (gside) DRV(gside->fork(&stw));   // after this layer's ln_act_bwd
      if (ln)
        DRV(colsum(g, grads->at(co.lnb[i]), E, B, hd, hd, hd, stw, nullptr, nullptr, dz, gb, g, xh, grads->at(co.lns[i])));
      else
        DRV(colsum(dz, gb, E, B, hd, hd, hd, stw));
      const float* a_ptr; long long a_zs, a_ks;
      if (i == 0) { a_ptr = inp; a_zs = 0; a_ks = ld_inp; }
      else { WSF(ws, st, hprev, (size_t)E * B * hp, "h%d_%c", i - 1, tag); a_ptr = hprev; a_zs = (long long)B * hp; a_ks = hp; }
      G gg; gg.a = a_ptr; gg.b = dz; gg.c = grads->at(co.w[i]); gg.nb = E; gg.M = hp; gg.N = hd; gg.K = B; gg.a_zs = a_zs;
      gg.a_ms = 1; gg.a_ks = a_ks; gg.b_zs = (long long)B * hd; gg.b_ks = hd; gg.b_ns = 1; gg.c_zs = (long long)hp * hd;
      gg.c_ms = hd; gg.a_kc = 0;
      DRV(gemm(gg, stw));
    }
    if (i > 0) {
      WSF(ws, st, dh, (size_t)E * B * hp, "edh%d", i - 1);
      G gg; gg.a = dz; gg.b = W; gg.c = dh; gg.nb = E; gg.M = B; gg.N = hp; gg.K = hd; gg.a_zs = (long long)B * hd; gg.a_ms = hd;
      gg.a_ks = 1; gg.b_zs = (long long)hp * hd; gg.b_ks = 1; gg.b_ns = hd; gg.c_zs = (long long)B * hp; gg.c_ms = hp; gg.b_nc = 0;
      DRV(gemm(gg, st));
      dy = dh;
    } else if (want_dinp == 0) {
      return SERL_OK;
    } else {
      WSF(ws, st, dinp, (size_t)B * hp, "dinp");
      const int ns = ksplit(E * hd);   // few (M, N) tiles and a long K = (member, feature) loop: split it
      G gg; gg.a = dz; gg.b = W; gg.M = B; gg.N = hp; gg.K = E * hd; gg.seglen = hd; gg.a_ss = (long long)B * hd; gg.a_ms = hd;
      gg.a_ks = 1; gg.b_ss = (long long)hp * hd; gg.b_ks = 1; gg.b_ns = hd; gg.c_ms = hp; gg.b_nc = 0;
      if (ns > 1) {
        WSF(ws, st, slab, (size_t)ns * B * hp, "dinp_slab");
        gg.c = slab; gg.nsplit = ns; gg.c_split = (long long)B * hp;
        DRV(gemm(gg, st));
        if (want_dinp == 2) { *dinp_out = slab; *nslab_out = ns; *ss_out = (long long)B * hp; return SERL_OK; }
        DRV(serl_ln_act_fwd(slab, ns, (long long)B * hp, hp, nullptr, nullptr, nullptr, 0, dinp, hp, nullptr, nullptr, B, hp,
                            B, 0, ACT_NONE, st));
      } else {
        gg.c = dinp;
        DRV(gemm(gg, st));
      }
      *dinp_out = dinp; *nslab_out = 1; *ss_out = 0;
    }
  }
  return SERL_OK;
}

// ================================================================= policy head (TanhNormal(MLP))
int policy_fwd(serl_learner* h, char tag, const Store& s, const ActorOff& ao, const float* z, int ldz, int B, Workspace& ws,
               hipStream_t st, const uint32_t* key, int row_offset, int total_rows, float* act, int ld_act,
               bool deterministic, bool save, bool want_logp, float* logp_out, float** logp_ret) {
  const Cfg& c = h->cfg;
  const int A = c.A;
  if (act && h->pol_fused && c.nh == 2 && c.hid[0] == 256 && c.hid[1] == 256 && A <= 8 && serl_get_matmul_precision() == 1) {
    // the whole pass in one launch (ensemble.hip, policy epilogue): three GEMM launches + the sample kernel less
    float *h0 = nullptr, *h1 = nullptr, *eps = nullptr, *logp = nullptr;
    if (save) {
      WSF(ws, st, a0, (size_t)B * 256, "ph0_%c", tag); h0 = a0;
      WSF(ws, st, a1, (size_t)B * 256, "ph1_%c", tag); h1 = a1;
      WSF(ws, st, e, (size_t)B * A, "eps_%c", tag); eps = e;
    }
    WSF(ws, st, ms, (size_t)B * 2 * A, "ms_%c", tag);
    if (want_logp) {
      if (logp_out) logp = logp_out;
      else { WSF(ws, st, l, (size_t)B, "logp_%c", tag); logp = l; }
    }
    DRV(serl_policy_fwd_fused(z, ldz, c.latent, B, s.at(ao.w[0]), s.at(ao.b[0]), s.at(ao.w[1]), s.at(ao.b[1]), s.at(ao.mk),
                              s.at(ao.lk), s.at(ao.mb), s.at(ao.lb), A, key ? key[0] : 0, key ? key[1] : 0, row_offset,
                              total_rows, deterministic, act, ld_act, ms, eps, logp, h0, h1, st));
    if (logp_ret) *logp_ret = logp;
    return SERL_OK;
  }
  const float* a_ptr = z; long long a_ms = ldz; int din = c.latent;
  for (int i = 0; i < c.nh; ++i) {
    const int hd = c.hid[i];
    WSF(ws, st, hbuf, (size_t)B * hd, "ph%d_%c", i, tag);
    G g; g.a = a_ptr; g.b = s.at(ao.w[i]); g.c = hbuf; g.bias = s.at(ao.b[i]); g.M = B; g.N = hd; g.K = din; g.a_ms = a_ms;
    g.a_ks = 1; g.b_ks = hd; g.b_ns = 1; g.c_ms = hd; g.relu = 1;
    DRV(gemm(g, st));
    a_ptr = hbuf; a_ms = hd; din = hd;
  }
  WSF(ws, st, ms, (size_t)B * 2 * A, "ms_%c", tag);
  {
    G g; g.a = a_ptr; g.b = s.at(ao.mk); g.c = ms; g.bias = s.at(ao.mb); g.nb = 2; g.M = B; g.N = A; g.K = din; g.a_zs = 0;
    g.a_ms = a_ms; g.a_ks = 1; g.b_zs = ao.lk - ao.mk; g.b_ks = A; g.b_ns = 1; g.c_zs = A; g.c_ms = 2 * A;
    g.bias_zs = ao.lb - ao.mb;
    DRV(gemm(g, st));
  }
  if (!act) {      // heads only (behaviour cloning: log-prob of given actions, no sample)
    if (logp_ret) *logp_ret = nullptr;
    return SERL_OK;
  }
  float* logp = nullptr;
  if (want_logp) {
    if (logp_out) logp = logp_out;
    else { WSF(ws, st, l, (size_t)B, "logp_%c", tag); logp = l; }
  }
  float* eps = nullptr;
  if (save) { WSF(ws, st, e, (size_t)B * A, "eps_%c", tag); eps = e; }
  DRV(serl_tanh_gaussian_fwd(ms, ms + A, 2 * A, act, ld_act, nullptr, eps, logp, B, A, key ? key[0] : 0, key ? key[1] : 0,
                             row_offset, total_rows, deterministic, st));
  if (logp_ret) *logp_ret = logp;
  return SERL_OK;
}

// backward from ws["dms"] (B, 2A) = [dmean | dlogstd]; *dz_out = (B, latent) when need_dz
int policy_bwd(serl_learner* h, char tag, const Store& s, const ActorOff& ao, const Store& gr, const float* z, int ldz,
               int B, Workspace& ws, hipStream_t st, bool need_dz, float** dz_out) {
  const Cfg& c = h->cfg;
  const int A = c.A, nl = c.nh;
  int dims[MAX_HID + 1];
  dims[0] = c.latent;
  for (int i = 0; i < nl; ++i) dims[i + 1] = c.hid[i];
  const int hl = dims[nl];
  WSF(ws, st, dms, (size_t)B * 2 * A, "dms");
  WSF(ws, st, hlast, (size_t)B * hl, "ph%d_%c", nl - 1, tag);
  {
    G g; g.a = hlast; g.b = dms; g.c = gr.at(ao.mk); g.nb = 2; g.M = hl; g.N = A; g.K = B; g.a_zs = 0; g.a_ms = 1; g.a_ks = hl;
    g.b_zs = A; g.b_ks = 2 * A; g.b_ns = 1; g.c_zs = ao.lk - ao.mk; g.c_ms = A; g.a_kc = 0;
    DRV(gemm(g, st));
  }
  DRV(colsum(dms, gr.at(ao.mb), 1, B, A, 2 * A, 0, st));
  DRV(colsum(dms + A, gr.at(ao.lb), 1, B, A, 2 * A, 0, st));
  WSF(ws, st, dh0, (size_t)B * hl, "pdh%d", nl - 1);
  {
    G g; g.a = dms; g.b = s.at(ao.mk); g.c = dh0; g.M = B; g.N = hl; g.K = 2 * A; g.seglen = A; g.a_ss = A; g.a_ms = 2 * A;
    g.a_ks = 1; g.b_ss = ao.lk - ao.mk; g.b_ks = 1; g.b_ns = A; g.c_ms = hl; g.b_nc = 0;
    DRV(gemm(g, st));
  }
  float* dh = dh0;
  for (int i = nl - 1; i >= 0; --i) {
    const int hd = dims[i + 1], hp = dims[i];
    WSF(ws, st, hbuf, (size_t)B * hd, "ph%d_%c", i, tag);
    WSF(ws, st, dz, (size_t)B * hd, "pdz%d", i);
    DRV(serl_ln_act_bwd(dh, hd, nullptr, nullptr, 0, hbuf, hd, nullptr, nullptr, nullptr, 0, dz, nullptr, B, hd, B, 0,
                        ACT_RELU, st));
    DRV(colsum(dz, gr.at(ao.b[i]), 1, B, hd, hd, 0, st));
    const float* a_ptr; long long a_ks;
    if (i == 0) { a_ptr = z; a_ks = ldz; }
    else { WSF(ws, st, hprev, (size_t)B * hp, "ph%d_%c", i - 1, tag); a_ptr = hprev; a_ks = hp; }
    {
      G g; g.a = a_ptr; g.b = dz; g.c = gr.at(ao.w[i]); g.M = hp; g.N = hd; g.K = B; g.a_ms = 1; g.a_ks = a_ks; g.b_ks = hd;
      g.b_ns = 1; g.c_ms = hd; g.a_kc = 0;
      DRV(gemm(g, st));
    }
    if (i > 0 || need_dz) {
      float* nd;
      if (i > 0) { WSF(ws, st, x, (size_t)B * hp, "pdh%d", i - 1); nd = x; }
      else { WSF(ws, st, x, (size_t)B * hp, "pdzlat"); nd = x; }
      G g; g.a = dz; g.b = s.at(ao.w[i]); g.c = nd; g.M = B; g.N = hp; g.K = hd; g.a_ms = hd; g.a_ks = 1; g.b_ks = 1;
      g.b_ns = hd; g.c_ms = hp; g.b_nc = 0;
      DRV(gemm(g, st));
      dh = nd;
    }
  }
  if (dz_out) *dz_out = need_dz ? dh : nullptr;
  return SERL_OK;
}

// ================================================================= feature stage of one or more passes
struct Feats {   // device inputs of the call, rows [0, n)
  const float* obs = nullptr; const float* next_obs = nullptr;         // state-based: (n, obs_dim)
  const float* state = nullptr; const float* next_state = nullptr;     // pixel-based: (n, state_dim)
  unsigned char* aug[2][MAX_CAM] = {};                                 // [obs|next][sorted cam] (n, H, W, 3*S) uint8
  float* bb[2][MAX_CAM] = {};                                          // mobilenet encoder: backbone features (n, fh, fw, fc)
  int row0 = 0, total_rows = 0;                                        // global row of this rank's minibatch row 0 / global minibatch
};
// which: 0 obs, 1 next; drop: rngs={"dropout": key} of an apply call with training=True (NULL: no dropout)
struct FPass { char tag; Store store; const EncOff* eo; int which; float* z; hipStream_t trunk_stream; const uint32_t* drop = nullptr;
               int lo = -1; };     // lo >= 0: this pass reads minibatch rows lo .. lo + B instead of the call's (merged run-ahead pass)

// TwoMobileNetEncoder behind its frozen backbone (two_mobilenet_encoder.py:33-34): SpatialLearnedEmbeddings of the
// minibatch rows of every pass (one grouped launch), then Dropout for the training passes.  The Dropout rng: flax folds
// the module path into rngs["dropout"] (not restatable); BUILD-DEFINED as fold_in(key, camera index), mask element =
// bernoulli bits of the (global minibatch, C*8) draw at the GLOBAL row (oracle/learner.py encoder_dropout).
int mobilenet_sle_fwd(serl_learner* h, const FPass* passes, int npass, const Feats& f, int lo, int B, Workspace& ws, hipStream_t st) {
  const Cfg& c = h->cfg;
  const int HW = c.fh * c.fw, F8 = c.fc * SLE_FEATURES;
  const void* feats[MAXG_]; const float* kers[MAXG_]; float* outs[MAXG_];
  int n = 0;
  SERL_CHECK_ARG(npass * c.ncam <= MAXG_, "learner: too many (pass, camera) groups");
  for (int i = 0; i < npass; ++i) {
    WSF(ws, st, enc, (size_t)B * c.enc_dim, "enc_%c", passes[i].tag);
    for (int ci = 0; ci < c.ncam; ++ci, ++n) {
      feats[n] = f.bb[passes[i].which][ci] + (size_t)lo * HW * c.fc;
      kers[n] = passes[i].store.at(passes[i].eo->sle[ci]);
      outs[n] = enc + (size_t)ci * F8;
    }
  }
  DRV(serl_sle_fwd(n, feats, kers, outs, c.enc_dim, B, HW, c.fc, SLE_FEATURES, 0, st));
  for (int i = 0; i < npass; ++i) {
    if (!passes[i].drop || c.enc_drop <= 0.f) continue;
    WSF(ws, st, enc, (size_t)B * c.enc_dim, "enc_%c", passes[i].tag);
    for (int ci = 0; ci < c.ncam; ++ci) {
      unsigned k[2];
      DRV(serl_prng_fold_in(passes[i].drop[0], passes[i].drop[1], (unsigned)ci, k));
      DRV(serl_dropout(enc + (size_t)ci * F8, c.enc_dim, B, F8, k[0], k[1], 1.0f - c.enc_drop, f.row0, f.total_rows, st));
    }
  }
  return SERL_OK;
}

int features(serl_learner* h, const FPass* passes, int npass, const Feats& f, int lo, int B, Workspace& ws, hipStream_t st,
             char save_tag, hipEvent_t fork, bool prep) {
  const Cfg& c = h->cfg;
  if (!c.pixel) {
    if (fork)
      for (int i = 0; i < npass; ++i) DRV(link(st, passes[i].trunk_stream, fork));
    for (int i = 0; i < npass; ++i) {
      const float* src = passes[i].which ? f.next_obs : f.obs;
      DRV(serl_copy2d(passes[i].z, c.ldz, src + (size_t)lo * c.obs_dim, c.obs_dim, B, c.obs_dim, passes[i].trunk_stream));
    }
    return SERL_OK;
  }
  if (c.encoder == 1) {
    DRV(mobilenet_sle_fwd(h, passes, npass, f, lo, B, ws, st));
  } else {
    const size_t img_bytes = (size_t)c.H * c.W * 3 * c.stack;
    Pass ep[4];
    for (int i = 0; i < npass; ++i) {
      ep[i].tag = passes[i].tag; ep[i].store = passes[i].store; ep[i].eo = passes[i].eo;
      const size_t plo = passes[i].lo >= 0 ? passes[i].lo : lo;
      for (int ci = 0; ci < c.ncam; ++ci) ep[i].imgs[ci] = f.aug[passes[i].which][ci] + plo * img_bytes;
    }
    DRV(encoder_fwd(h, ep, npass, B, ws, st, save_tag, prep));
  }
  if (fork)     // the trunks of the passes are independent: side streams wait for the encoders
    for (int i = 0; i < npass; ++i) DRV(link(st, passes[i].trunk_stream, fork));
  for (int i = 0; i < npass; ++i) {
    const float* state = (passes[i].which ? f.next_state : f.state) + (size_t)(passes[i].lo >= 0 ? passes[i].lo : lo) * c.state_dim;
    DRV(trunk_fwd(h, passes[i].tag, passes[i].store, *passes[i].eo, state, passes[i].z, c.ldz, B, ws, passes[i].trunk_stream,
                  passes[i].tag == save_tag));
  }
  return SERL_OK;
}

int features_bwd(serl_learner* h, char tag, const Store& store, const EncOff& eo, const Store& grads, const Feats& f, int lo,
                 int B, Workspace& ws, hipStream_t st, const float* dz, int lddz, bool need_encoder, GradSide* gside,
                 int dz_nslab, long long dz_ss, const uint32_t* drop = nullptr,
                 const std::function<int()>* after_trunk = nullptr, GradSide* gside_trunk = nullptr) {
  const Cfg& c = h->cfg;
  if (!c.pixel) return SERL_OK;   // no trainable feature stage for state inputs
  WSF(ws, st, z, (size_t)B * c.ldz, "z%c", tag);
  DRV(trunk_bwd(h, tag, store, eo, grads, f.state + (size_t)lo * c.state_dim, z, c.ldz, dz, lddz, B, ws, st, need_encoder,
                gside_trunk ? gside_trunk : gside, dz_nslab, dz_ss));
  // every gradient kernel of the ensemble and trunk leaves is enqueued: the data-parallel bucket of those leaves starts here
  if (after_trunk) DRV((*after_trunk)());
  if (need_encoder && c.encoder == 1) {
    // frozen backbone: the gradient stops at its feature map; only the spatial-embedding kernels are trained.  Backward
    // of the Dropout = the same mask on the gradient, then dK[h,w,c,f] = sum_b feat[b,h,w,c] denc[b,c,f].
    const int HW = c.fh * c.fw, F8 = c.fc * SLE_FEATURES;
    WSF(ws, st, denc, (size_t)B * c.enc_dim, "denc_%c", tag);
    const float* douts[MAX_CAM]; const float* kers[MAX_CAM]; const void* feats[MAX_CAM]; float* dker[MAX_CAM];
    for (int ci = 0; ci < c.ncam; ++ci) {
      if (drop && c.enc_drop > 0.f) {
        unsigned k[2];
        DRV(serl_prng_fold_in(drop[0], drop[1], (unsigned)ci, k));
        DRV(serl_dropout(denc + (size_t)ci * F8, c.enc_dim, B, F8, k[0], k[1], 1.0f - c.enc_drop, f.row0, f.total_rows, st));
      }
      douts[ci] = denc + (size_t)ci * F8;
      kers[ci] = store.at(eo.sle[ci]);
      feats[ci] = f.bb[0][ci] + (size_t)lo * HW * c.fc;
      dker[ci] = grads.at(eo.sle[ci]);
    }
    hipStream_t sw = st;
    if (gside) DRV(gside->fork(&sw));
    DRV(serl_sle_bwd(c.ncam, douts, c.enc_dim, kers, feats, nullptr, dker, B, HW, c.fc, SLE_FEATURES, 0, sw));
  } else if (need_encoder) {
    const size_t img_bytes = (size_t)c.H * c.W * 3 * c.stack;
    const unsigned char* imgs[MAX_CAM];
    for (int ci = 0; ci < c.ncam; ++ci) imgs[ci] = f.aug[0][ci] + lo * img_bytes;
    DRV(encoder_bwd(h, tag, store, eo, grads, imgs, B, ws, st, gside, true));
  }
  return SERL_OK;
}

// Fused frame-stack gather + _unpack + random shift of all n rows of one side (0 obs, 1 next_obs), one kernel per camera
// (serl/utils/augmentations.py:5-11,44-48; drq_learner.py:131-137).  Crop offsets are keyed by the GLOBAL row index so that
// rank shards reproduce the single-process stream; keys in the caller's pixel_keys order: key, rng = split(rng)
// per camera.
int augment_side(serl_learner* h, const serl_pixel_source* pixels, int side, const uint32_t key_side[2], int n, int utd,
                 int B, int rank, int world, Workspace& ws, hipStream_t st, Feats& f) {
  const Cfg& c = h->cfg;
  const int S = c.stack, pad = 4;   // batched_random_crop(..., padding=4), serl/utils/augmentations.py:44
  const size_t img = (size_t)c.H * c.W * 3 * S;
  uint32_t cam_key[MAX_CAM][2], kk[2] = {key_side[0], key_side[1]};
  for (int p = 0; p < c.ncam; ++p) {
    uint32_t k[2], r[2];
    split2(kk, k, r);
    cam_key[p][0] = k[0]; cam_key[p][1] = k[1];
    kk[0] = r[0]; kk[1] = r[1];
  }
  for (int ci = 0; ci < c.ncam; ++ci) {
    const uint32_t* key = cam_key[c.aug_rank[ci]];
    WSB(ws, st, int, off, (size_t)n * 2 * 4, "off_%d_%d", side, ci);
    if (world == 1) {
      DRV(serl_crop_offsets(key[0], key[1], n, 0, n, pad, off, st));
    } else {
      for (int i = 0; i < utd; ++i)
        DRV(serl_crop_offsets(key[0], key[1], B, i * B * world + rank * B, n * world, pad, off + 2 * i * B, st));
    }
    WSB(ws, st, unsigned char, dst, (size_t)n * img, "aug_%d_%d", side, ci);
    f.aug[side][ci] = dst;
    const serl_pixel_source& src = pixels[ci];
    if (src.kind == 0) {          // device frame store + window indices
      SERL_CHECK_ARG(src.frames && src.indx, "learner_update: frame store source without frames / indices");
      DRV(serl_gather_crop_frames(src.frames, src.capacity, src.indx, S + 1, side, off, dst, n, c.H, c.W, S, pad, st));
    } else if (src.kind == 1) {   // packed (n,H,W,3,S+1)
      SERL_CHECK_ARG(src.obs, "learner_update: packed source missing");
      DRV(serl_gather_crop_packed(src.obs, S + 1, side, off, dst, n, c.H, c.W, S, pad, st));
    } else {                      // unpacked: obs (n,H,W,3,S) and next_obs (n,H,W,3,S)
      const unsigned char* p = side ? src.next_obs : src.obs;
      SERL_CHECK_ARG(p, "learner_update: unpacked source missing");
      DRV(serl_gather_crop_packed(p, S, 0, off, dst, n, c.H, c.W, S, pad, st));
    }
  }
  return SERL_OK;
}

// The backbone is frozen and the same in every network of the learner (a module attribute, not a parameter:
// two_mobilenet_encoder.py:12,31), so its features of the augmented images of one side are computed ONCE per call for all
// n rows; the passes of every step only pool them (trainable SLE kernels).
int backbone_side(serl_learner* h, int side, int n, Workspace& ws, hipStream_t st, Feats& f) {
  const Cfg& c = h->cfg;
  SERL_CHECK_ARG(h->backbone, "learner_update: serl_learner_set_backbone has not been called");
  // images per launch chain: the layers behind the third stride have 64..16 rows per image, and at 256 images their
  // launches are latency-bound (1.85 ms per 256 images; 512: 1.53, 1024: 1.43, 2048: 1.40); 1.9 MB of workspace per image
  const int chunk = std::min(n, 1024);
  const long long wsb = serl_mbv3_workspace_bytes(chunk, c.H, c.W);
  WSF(ws, st, bbws, (size_t)wsb / 4, "bb_ws");
  const size_t fsz = (size_t)c.fh * c.fw * c.fc, img1 = (size_t)c.H * c.W * 3;
  for (int ci = 0; ci < c.ncam; ++ci) {
    WSF(ws, st, bb, (size_t)n * fsz, "bb_%d_%d", side, ci);
    f.bb[side][ci] = bb;
    for (int r0 = 0; r0 < n; r0 += chunk) {
      const int m = std::min(chunk, n - r0);
      DRV(serl_mbv3_forward(h->backbone, f.aug[side][ci] + r0 * img1, 1, m, c.H, c.W, bbws, wsb, bb + r0 * fsz, st));
    }
  }
  return SERL_OK;
}

// REDQ subsample (serl/networks/ensemble.py:25-37) -> first member / member stride when the picked members form an
// arithmetic sequence (always for 2 of E), else a gathered copy of the members in a scratch store
int target_members(serl_learner* h, const uint32_t key[2], const Store& target, Workspace& ws, hipStream_t st, int* m0,
                   int* mstride, int* emin, Store* tstore) {
  const Cfg& c = h->cfg;
  *tstore = target;
  if (c.num_min_qs <= 0) { *m0 = 0; *mstride = 1; *emin = c.E; return SERL_OK; }
  int idx[64];
  SERL_CHECK_ARG(c.num_min_qs <= 64, "learner: num_min_qs > 64");
  DRV(serl_prng_choice(key[0], key[1], c.E, c.num_min_qs, idx));
  const int m = c.num_min_qs;
  *emin = m;
  if (m == 1) { *m0 = idx[0]; *mstride = 1; return SERL_OK; }
  const int d = idx[1] - idx[0];
  bool arith = true;
  for (int i = 0; i + 1 < m; ++i) arith = arith && (idx[i + 1] - idx[i] == d);
  if (arith) { *m0 = idx[0]; *mstride = d; return SERL_OK; }
  // general num_min_qs > 2: copy the chosen members of the head leaves into a scratch store (rare path)
  const CriticOff& co = h->co;
  WSF(ws, st, scratch, (size_t)co.size, "tgt_scratch");
  int dims[MAX_HID + 1];
  dims[0] = c.latent + c.crit_A;
  for (int i = 0; i < c.nh; ++i) dims[i + 1] = c.hid[i];
  auto gather = [&](long long off, long long per) -> int {
    for (int j = 0; j < m; ++j)
      DRV(serl_copy2d(scratch + off + j * per, (int)per, target.base + off + idx[j] * per, (int)per, 1, (int)per, st));
    return SERL_OK;
  };
  for (int i = 0; i < c.nh; ++i) {
    DRV(gather(co.w[i], (long long)dims[i] * dims[i + 1]));
    DRV(gather(co.b[i], dims[i + 1]));
    if (c.ln) { DRV(gather(co.lns[i], dims[i + 1])); DRV(gather(co.lnb[i], dims[i + 1])); }
  }
  DRV(gather(co.hk, dims[c.nh]));
  DRV(gather(co.hb, 1));
  tstore->base = scratch;
  *m0 = 0; *mstride = 1;
  return SERL_OK;
}

}  // namespace

// ================================================================= C ABI
extern "C" int serl_learner_create(const serl_learner_config* pc, const long long* critic_offsets, int n_critic,
                                   const long long* actor_offsets, int n_actor, serl_learner** out) {
  SERL_CHECK_ARG(pc && critic_offsets && actor_offsets && out, "learner_create: null pointer");
  SERL_CHECK_ARG(pc->num_hidden >= 1 && pc->num_hidden <= MAX_HID, "learner_create: 1..%d hidden layers", MAX_HID);
  SERL_CHECK_ARG(pc->num_qs >= 1 && pc->action_dim >= 1, "learner_create: bad ensemble / action size");
  SERL_CHECK_ARG(!pc->pixel || (pc->num_cameras >= 1 && pc->num_cameras <= MAX_CAM &&
                                (pc->encoder == 1 || (pc->encoder == 0 && pc->num_conv >= 1 && pc->num_conv <= MAX_CONV))),
                 "learner_create: 1..%d cameras, 1..%d conv layers", MAX_CAM, MAX_CONV);
  SERL_CHECK_ARG(pc->critic_dropout >= 0.f && pc->critic_dropout < 1.f, "learner_create: critic_dropout must be in [0, 1)");
  serl_learner* h = new serl_learner();
  Cfg& c = h->cfg;
  memset(&c, 0, sizeof(c));
  c.A = pc->action_dim; c.nh = pc->num_hidden; c.E = pc->num_qs; c.num_min_qs = pc->num_min_qs;
  for (int i = 0; i < c.nh; ++i) c.hid[i] = pc->hidden_dims[i];
  c.ln = pc->critic_layer_norm; c.backup_entropy = pc->backup_entropy; c.has_temp = pc->has_temperature;
  c.pixel = pc->pixel; c.obs_dim = pc->obs_dim; c.ncam = pc->num_cameras; c.H = pc->height; c.W = pc->width;
  c.stack = pc->stack; c.state_dim = pc->state_dim; c.latent_dim = pc->latent_dim; c.nl = pc->num_conv;
  c.discount = pc->discount; c.tau = pc->tau; c.target_entropy = pc->target_entropy;
  c.actor_lr = pc->actor_lr; c.critic_lr = pc->critic_lr; c.temp_lr = pc->temp_lr;
  c.encoder = pc->encoder; c.fh = pc->feat_h; c.fw = pc->feat_w; c.fc = pc->feat_c; c.enc_drop = pc->encoder_dropout;
  c.crit_drop = pc->critic_dropout;
  if (c.pixel && c.encoder == 1) {
    int fh, fw, fc;
    const bool ok = c.stack == 1 && serl_mbv3_feature_shape(c.H, c.W, &fh, &fw, &fc) == SERL_OK && fh == c.fh && fw == c.fw &&
                    fc == c.fc && c.enc_drop >= 0.f && c.enc_drop < 1.f;
    if (!ok) {
      delete h;
      SERL_CHECK_ARG(false, "learner_create: mobilenet encoder needs (H, W, 3, 1) images of at least 32x32 and the backbone's feature shape");
    }
    c.nl = 0;
    for (int ci = 0; ci < c.ncam; ++ci) c.aug_rank[ci] = pc->aug_rank[ci];
    c.enc_dim = c.ncam * c.fc * SLE_FEATURES;
    c.latent = TRUNK_DIM + c.latent_dim;
  } else if (c.pixel) {
    int hh = c.H, ww = c.W, cc = 3 * c.stack;
    for (int li = 0; li < c.nl; ++li) {
      const int f = pc->cnn_features[li], k = pc->cnn_filters[li], s = pc->cnn_strides[li];
      if (k < 1 || s < 1 || f < 1 || hh < k || ww < k) {
        delete h;
        SERL_CHECK_ARG(false, "learner_create: conv layer %d does not fit its input", li);
      }
      ConvL& L = c.L[li];
      L.IH = hh; L.IW = ww; L.C = cc; L.K = k; L.S = s; L.F = f; L.OH = (hh - k) / s + 1; L.OW = (ww - k) / s + 1;
      hh = L.OH; ww = L.OW; cc = f;
    }
    for (int ci = 0; ci < c.ncam; ++ci) c.aug_rank[ci] = pc->aug_rank[ci];
    c.enc_dim = c.ncam * c.L[c.nl - 1].F * SLE_FEATURES;
    c.latent = TRUNK_DIM + c.latent_dim;
  } else {
    c.latent = c.obs_dim;
  }
  c.ldz = c.latent + c.A;
  c.crit_A = pc->critic_no_action ? 0 : c.A;
  // parameter leaf offsets, in the order serl_amd/networks/spec.py lays the stores out
  auto enc = [&](const long long* o, int n, int* cur, EncOff* e) -> bool {
    const int need = c.ncam * c.nl * 2 + c.ncam + 8;
    if (*cur + need > n) return false;
    for (int ci = 0; ci < c.ncam; ++ci)
      for (int li = 0; li < c.nl; ++li) { e->ck[ci][li] = o[(*cur)++]; e->cb[ci][li] = o[(*cur)++]; }
    for (int ci = 0; ci < c.ncam; ++ci) e->sle[ci] = o[(*cur)++];
    e->d0k = o[(*cur)++]; e->d0b = o[(*cur)++]; e->l0s = o[(*cur)++]; e->l0b = o[(*cur)++];
    e->d1k = o[(*cur)++]; e->d1b = o[(*cur)++]; e->l1s = o[(*cur)++]; e->l1b = o[(*cur)++];
    return true;
  };
  bool ok = true;
  int cur = 0;
  memset(&h->co, 0, sizeof(h->co));
  memset(&h->ao, 0, sizeof(h->ao));
  if (c.pixel) ok = ok && enc(critic_offsets, n_critic, &cur, &h->co.e);
  const int per = c.ln ? 4 : 2;
  ok = ok && cur + c.nh * per + 3 == n_critic;
  if (ok) {
    for (int i = 0; i < c.nh; ++i) {
      h->co.w[i] = critic_offsets[cur++]; h->co.b[i] = critic_offsets[cur++];
      if (c.ln) { h->co.lns[i] = critic_offsets[cur++]; h->co.lnb[i] = critic_offsets[cur++]; }
    }
    h->co.hk = critic_offsets[cur++]; h->co.hb = critic_offsets[cur++];
    h->co.size = critic_offsets[cur++];   // last entry: total size of the store in floats
  }
  cur = 0;
  if (ok && c.pixel) ok = enc(actor_offsets, n_actor, &cur, &h->ao.e);
  ok = ok && cur + c.nh * 2 + 5 == n_actor;
  if (ok) {
    for (int i = 0; i < c.nh; ++i) { h->ao.w[i] = actor_offsets[cur++]; h->ao.b[i] = actor_offsets[cur++]; }
    h->ao.mk = actor_offsets[cur++]; h->ao.lk = actor_offsets[cur++]; h->ao.mb = actor_offsets[cur++];
    h->ao.lb = actor_offsets[cur++];
    h->ao.size = actor_offsets[cur++];
  }
  if (!ok) {
    delete h;
    SERL_CHECK_ARG(false, "learner_create: offset tables do not match the network description (%d critic, %d actor entries)",
                   n_critic, n_actor);
  }
  hipError_t e = hipGetDevice(&h->device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->s1, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->astream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->gside.own, hipStreamNonBlocking);
  // (only when asked for: a process has GPU_MAX_HW_QUEUES = 4 hardware queues, a fifth stream shares one with another)
  if (e == hipSuccess && h->wg_streams > 1) e = hipStreamCreateWithFlags(&h->gside2.own, hipStreamNonBlocking);
  if (e == hipSuccess && h->actor_cus < 0) {
    int ncu = 0;
    e = hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device);
    h->actor_cus = ncu * 7 / 8;
  }
  hipEvent_t* evs[] = {&h->fork_ev, &h->join1, &h->astart, &h->agate, &h->aobs};
  for (hipEvent_t* ev : evs)
    if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
  if (e != hipSuccess) {
    serl_set_error("learner_create: %s", hipGetErrorString(e));
    delete h;
    return SERL_ERR_HIP;
  }
  *out = h;
  return SERL_OK;
}

// frozen MobileNetV3 parameters (serl_mbv3_param_floats floats, device) shared by every network of the learner
extern "C" int serl_learner_set_backbone(serl_learner* h, const float* params) {
  SERL_CHECK_ARG(h && params, "learner_set_backbone: null pointer");
  h->backbone = params;
  return SERL_OK;
}

extern "C" int serl_learner_destroy(serl_learner* h) {
  if (!h) return SERL_OK;
  if (g_hp.on && g_hp.n_upd > 0) {
    fprintf(stderr, "[host prof] %ld update calls: %.3f ms each; per call: %.1f event records %.3f ms, %.1f stream waits %.3f ms, "
            "%.0f workspace lookups\n", g_hp.n_upd, g_hp.t_upd / g_hp.n_upd * 1e3, (double)g_hp.n_rec / g_hp.n_upd,
            g_hp.t_rec / g_hp.n_upd * 1e3, (double)g_hp.n_wait / g_hp.n_upd, g_hp.t_wait / g_hp.n_upd * 1e3,
            (double)g_hp.n_ws / g_hp.n_upd);
    g_hp = HostProf{};
  }
  (void)hipDeviceSynchronize();
  for (auto& kv : h->ws_update) kv.second.release();
  h->ws_act.release();
  if (h->s1) (void)hipStreamDestroy(h->s1);
  if (h->astream) (void)hipStreamDestroy(h->astream);
  if (h->gside.own) (void)hipStreamDestroy(h->gside.own);
  for (hipEvent_t e : h->gside.ev) ev_destroy(e);
  if (h->comm) (void)hipStreamDestroy(h->comm);
  for (hipEvent_t e : {h->comm_fork, h->comm_done}) ev_destroy(e);
  if (h->gside2.own) (void)hipStreamDestroy(h->gside2.own);
  for (hipEvent_t e : h->gside2.ev) ev_destroy(e);
  for (hipEvent_t e : h->adone) ev_destroy(e);
  for (hipEvent_t e : {h->fork_ev, h->join1, h->astart, h->agate, h->aobs}) ev_destroy(e);
  for (auto& r : h->trecs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto& r : h->precs) (void)hipEventDestroy(r.ev);
  delete h;
  return SERL_OK;
}

// One whole update(batch, utd_ratio): see the file header.  Everything is enqueued; nothing waits for the GPU.
extern "C" int serl_learner_update(serl_learner* h, const serl_agent_buffers* ag, const serl_update_batch* bt, int utd,
                                   const unsigned* rng_in, unsigned* rng_out, float* info_out, void* stream,
                                   serl_allreduce_fn allreduce, void* allreduce_ctx, int rank, int world) {
  SERL_CHECK_ARG(h && ag && bt && rng_in && rng_out && info_out, "learner_update: null pointer");
  struct UpdTimer { double t0 = g_hp.on ? HostProf::now() : 0; ~UpdTimer() { if (g_hp.on) { g_hp.t_upd += HostProf::now() - t0; ++g_hp.n_upd; } } } upd_timer;
  SERL_CHECK_ARG(utd >= 1 && bt->n >= utd && bt->n % utd == 0, "learner_update: batch of %d rows is not divisible by utd_ratio %d", bt->n, utd);
  SERL_CHECK_ARG(world >= 1 && rank >= 0 && rank < world && (world == 1 || allreduce), "learner_update: bad rank / world / allreduce");
  const Cfg& c = h->cfg;
  const CriticOff& co = h->co;
  const ActorOff& ao = h->ao;
  const int n = bt->n, B = n / utd, A = c.A, ldz = c.ldz, E = c.E, Bg = B * world;
  // Pixel learners spread a call over four streams (file header).  State-based learners are a string of ~20 small,
  // latency-bound launches per critic step: there a cross-stream dependency (hipEventRecord + hipStreamWaitEvent, ~10-20 us
  // of host time each) costs more than the overlap returns, so everything runs on the caller's stream.
  const bool multi = c.pixel != 0;
  hipStream_t st = (hipStream_t)stream, s1 = multi ? h->s1 : st, sta = multi ? h->astream : st;
  Workspace& ws = h->ws_update[B];
  const Store actor{ag->actor}, critic{ag->critic}, target{ag->target};
  const Store cgr{ag->critic_grads}, agr{ag->actor_grads};
  float* log_temp = ag->temp;

  // ---- rng chain of the call.  aug_mode 1 (DrQLearner.update, drq_learner.py:174-177):
  // rng, key = split(rng) twice (rng is child 0).  aug_mode 2 (PixelDDPGLearner.update, pixel_ddpg_learner.py:167-180):
  // key, rng = split(rng) twice, and the update itself runs on the rng the augmentation did NOT advance.
  uint32_t rng[2] = {rng_in[0], rng_in[1]}, key_obs[2] = {0, 0}, key_next[2] = {0, 0};
  if (c.pixel) {
    uint32_t a[2], b[2];
    if (bt->aug_mode == 2) {
      split2(rng, key_obs, a);
      split2(a, key_next, b);
    } else {
      split2(rng, a, key_obs);
      split2(a, b, key_next);
      rng[0] = b[0]; rng[1] = b[1];
    }
  }
  if (bt->use_sac_rng) { rng[0] = bt->sac_rng[0]; rng[1] = bt->sac_rng[1]; }
  std::vector<uint32_t> step_keys((size_t)utd * 8);   // per step: next-action noise, subsample, target dropout, critic dropout
  for (int i = 0; i < utd; ++i) {
    uint32_t k[2], r[2];
    uint32_t* sk = &step_keys[8 * i];
    split2(rng, k, r); sk[0] = k[0]; sk[1] = k[1];         // next-action noise
    split2(r, k, rng); sk[2] = k[0]; sk[3] = k[1];         // subsample_ensemble
    split2(rng, k, r); sk[4] = k[0]; sk[5] = k[1];         // dropout key of the target apply (only the MobileNet encoder has a Dropout)
    split2(r, k, rng); sk[6] = k[0]; sk[7] = k[1];         // dropout key of the critic apply
  }
  uint32_t key_pi[2], key_drop_pi[2], r2[2];
  split2(rng, key_pi, r2);            // actor step (sac_learner.py:121-122): policy sample key
  split2(r2, key_drop_pi, rng);       // ... and its critic dropout key
  rng_out[0] = rng[0]; rng_out[1] = rng[1];

  // ---- inputs
  Feats f;
  f.row0 = rank * B; f.total_rows = Bg;
  f.obs = bt->obs; f.next_obs = bt->next_obs; f.state = bt->state; f.next_state = bt->next_state;
  SERL_CHECK_ARG(bt->actions && bt->rewards && bt->masks, "learner_update: actions / rewards / masks missing");
  if (!c.pixel) SERL_CHECK_ARG(f.obs && f.next_obs, "learner_update: observations missing");
  float* info = info_out;   // 8 floats of the caller: the data-parallel hook all-reduces its first 4
  if (c.pixel) {
    SERL_CHECK_ARG(f.state && f.next_state, "learner_update: state leaves missing");
    for (int side = 0; side < 2; ++side)
      DRV(augment_side(h, bt->pixels, side, side ? key_next : key_obs, n, utd, B, rank, world, ws, st, f));
    if (c.encoder == 1)
      for (int side = 0; side < 2; ++side) DRV(backbone_side(h, side, n, ws, st, f));
  }

  if (bt->features_hook) {
    SERL_CHECK_ARG(allreduce, "learner_update: features_hook without a hook function");
    DRV(allreduce(allreduce_ctx, 3));
  }
  WSF(ws, st, zA, (size_t)B * ldz, "zA");
  WSF(ws, st, zC1, (size_t)B * ldz, "zC");
  WSF(ws, st, zP, (size_t)B * ldz, "zC_pi");                    // critic input of the actor step
  WSF(ws, st, zT_all, (size_t)utd * B * ldz, "zT_all");         // per-step inputs of the target ensemble
  WSF(ws, st, nlogp_all, (size_t)utd * B, "nlogp_all");
  WSF(ws, st, dq, (size_t)E * B, "dq");
  WSF(ws, st, target_q, (size_t)B, "target_q");
  WSF(ws, st, logpA, (size_t)B, "logp_A");
  while ((int)h->adone.size() < utd) {
    hipEvent_t e;
    DRV_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h->adone.push_back(e);
  }

  // ---- the actor's no-grad pass over next_obs runs one step ahead on its own stream (constant parameters in the loop)
  DRV(link(st, sta, h->astart));
  // State-based learners: the "feature stage" is a copy of the observation rows.  The policy reads them in place, and the
  // critic / target inputs [obs | action] of ALL utd minibatches are assembled by three launches here instead of four
  // per critic step (every launch costs ~10 us of a ~200 us step on this latency-bound chain).
  float* zC_all = nullptr;
  if (!c.pixel) {
    WSF(ws, st, zc, (size_t)n * ldz, "zC_all");
    zC_all = zc;
    DRV(serl_copy2d(zT_all, ldz, f.next_obs, c.obs_dim, n, c.obs_dim, st));
    DRV(serl_copy2d(zC_all, ldz, f.obs, c.obs_dim, n, c.obs_dim, st));
    DRV(serl_copy2d(zC_all + c.latent, ldz, bt->actions, A, n, A, st));
  }
  auto enqueue_actor = [&](int j) -> int {
    if (!c.pixel)
      return policy_fwd(h, 'A', actor, ao, f.next_obs + (size_t)j * B * c.obs_dim, c.obs_dim, B, ws, sta, &step_keys[8 * j],
                        rank * B, Bg, zT_all + ((size_t)j * B * ldz + c.latent), ldz, false, false, c.backup_entropy != 0,
                        nlogp_all + (size_t)j * B, nullptr);
    FPass p{'A', actor, &ao.e, 1, zA, sta};
    if (sta != st) (void)serl_convs_set_cu_limit(h->actor_cus);
    const int fe = features(h, &p, 1, f, j * B, B, ws, sta, 0, nullptr, j == 0);   // S-format weight copies made once (j == 0)
    (void)serl_convs_set_cu_limit(0);
    DRV(fe);
    DRV(policy_fwd(h, 'A', actor, ao, zA, ldz, B, ws, sta, &step_keys[8 * j], rank * B, Bg,
                   zT_all + ((size_t)j * B * ldz + c.latent), ldz, false, false, c.backup_entropy != 0,
                   nlogp_all + (size_t)j * B, nullptr));
    if (sta != st) DRV_HIP(ev_record(h->adone[j], sta));
    return SERL_OK;
  };
  auto enqueue_actor_obs = [&]() -> int {   // behind the last pass: the actor's own pass over obs for the actor step
    if (!c.pixel)
      return policy_fwd(h, 'A', actor, ao, f.obs + (size_t)(utd - 1) * B * c.obs_dim, c.obs_dim, B, ws, sta, key_pi, rank * B,
                        Bg, zP + c.latent, ldz, false, true, true, logpA, nullptr);
    FPass p{'A', actor, &ao.e, 0, zA, sta};
    if (sta != st) (void)serl_convs_set_cu_limit(h->actor_cus);
    const int fe = features(h, &p, 1, f, (utd - 1) * B, B, ws, sta, 'A', nullptr, false);
    (void)serl_convs_set_cu_limit(0);
    DRV(fe);
    DRV(policy_fwd(h, 'A', actor, ao, zA, ldz, B, ws, sta, key_pi, rank * B, Bg, zP + c.latent, ldz, false, true, true, logpA,
                   nullptr));
    if (sta != st) DRV_HIP(ev_record(h->aobs, sta));
    return SERL_OK;
  };
  DRV(enqueue_actor(0));

  int critic_step = ag->critic_step;
  for (int i = 0; i < utd; ++i) {
    const int lo = i * B;
    DRV(phase(h, "step", st));
    // ---- critic step (sac_learner.py:160-219)
    int m0, mstride, emin;
    Store tstore;
    DRV(target_members(h, &step_keys[8 * i + 2], target, ws, st, &m0, &mstride, &emin, &tstore));
    float* zT = zT_all + (size_t)i * B * ldz;
    float* zC = c.pixel ? zC1 : zC_all + (size_t)lo * ldz;
    // SERL_AMD_ACTOR_MERGE=1 (experiment): the actor's run-ahead ENCODER pass of step i + 1 rides in the target + critic
    // launches of step i as a third pass (6 groups per launch) instead of running next to the heads / backward phases
    const bool merge_actor = h->actor_merge && c.pixel && c.encoder == 0 && sta != st && i + 1 < utd;
    if (c.pixel) {
      // target chain on the main stream, critic chain on a side stream, next action from the actor stream
      FPass ps[3] = {{'T', target, &co.e, 1, zT, st, &step_keys[8 * i + 4]}, {'C', critic, &co.e, 0, zC, s1, &step_keys[8 * i + 6]},
                     {'A', actor, &ao.e, 1, zA, sta, nullptr, (i + 1) * B}};
      if (merge_actor) DRV_HIP(ev_wait(st, h->adone[i]));      // pass i's trunk has read enc_A before this launch overwrites it
      DRV(features(h, ps, merge_actor ? 3 : 2, f, lo, B, ws, st, 'C', h->fork_ev, true));
    }
    DRV(phase(h, "features_fwd", st));
    if (merge_actor) {
      const int j = i + 1;
      DRV(policy_fwd(h, 'A', actor, ao, zA, ldz, B, ws, sta, &step_keys[8 * j], rank * B, Bg,
                     zT_all + ((size_t)j * B * ldz + c.latent), ldz, false, false, c.backup_entropy != 0,
                     nlogp_all + (size_t)j * B, nullptr));
      DRV_HIP(ev_record(h->adone[j], sta));
    } else {
      // start pass i+1 when the main stream leaves its encoder forward (round 3, re-measured with the fused kernels: behind
      // the TD loss instead -1.3 %, in front of the encoder forward -16 % -- two persistent grids then share every CU's LDS)
      DRV(link(st, sta, h->agate));
      if (i + 1 < utd) DRV(enqueue_actor(i + 1));
      else DRV(enqueue_actor_obs());
      if (sta != st) DRV_HIP(ev_wait(st, h->adone[i]));
    }
    float *next_q, *q;
    if (!multi && c.crit_drop <= 0.f && ens_fwd_fusable(h)) {
      // state-based learners: target and critic passes share the caller's stream -> one launch for both
      serl_ens_fwd_job jt, jc;
      DRV(ens_fwd_job(h, 'T', tstore, co, m0, mstride, emin, zT, ldz, B, ws, st, false, &jt));
      DRV(ens_fwd_job(h, 'C', critic, co, 0, 1, E, zC, ldz, B, ws, st, true, &jc));
      DRV(serl_ensemble_fwd_fused_pair(&jt, &jc, st));
      next_q = jt.q; q = jc.q;
    } else {
      DRV(ensemble_fwd(h, 'T', tstore, co, m0, mstride, emin, zT, ldz, B, ws, st, false, &next_q, &step_keys[8 * i + 4], rank * B, Bg));
      if (c.pixel) DRV(serl_copy2d(zC + c.latent, ldz, bt->actions + (size_t)lo * A, A, B, A, s1));
      DRV(ensemble_fwd(h, 'C', critic, co, 0, 1, E, zC, ldz, B, ws, s1, true, &q, &step_keys[8 * i + 6], rank * B, Bg));
    }
    DRV(link(s1, st, h->join1));
    DRV(serl_td_loss(next_q, emin, bt->rewards + lo, bt->masks + lo, c.backup_entropy ? nlogp_all + (size_t)i * B : nullptr,
                     log_temp, c.discount, q, E, B, 1.0f / ((float)E * Bg), target_q, dq, info, st));
    DRV(phase(h, "heads_fwd", st));
    h->gside.begin(st, multi);
    h->gside2.begin(st, multi && h->wg_streams > 1);
    // pixel learners: the trunk backward sums the split-K slabs of the input gradient on read; state-based:
    // observations are not trainable, the critic step needs no input gradient at all
    float* dinp = nullptr; int nslab = 1; long long ss = 0;
    // SERL_AMD_WG_STREAMS=3: the heads' parameter gradients (ensemble, trunk) on a side stream of their own, so that the conv
    // weight gradients start when their dz is ready instead of behind ~200 us of small GEMMs
    GradSide* const gs_heads = multi && h->wg_streams == 3 ? &h->gside2 : &h->gside;
    DRV(ensemble_bwd(h, 'C', critic, co, &cgr, dq, zC, ldz, B, ws, st, gs_heads, c.pixel ? 2 : 0, &dinp, &nslab, &ss,
                     &step_keys[8 * i + 6], rank * B, Bg));
    DRV(phase(h, "ensemble_bwd", st));
    // data parallel, pixel learners: two buckets.  Bucket 4 = the trunk + ensemble leaves (the suffix of the flat gradient:
    // Dense(4096 -> 512), LayerNorms, 10 critics = 17 of 21 MB), all-reduced on the comm stream behind their gradient
    // kernels and NEXT TO the encoder backward; bucket 5 = the encoder leaves, on the main stream once the last weight
    // gradient is in.  The hook enqueues bucket 4 on the stream serl_learner_comm_stream returns.
    const bool dp_overlap = world > 1 && multi && h->dp_overlap;
    const std::function<int()> bucket4 = [&]() -> int {
      DRV(link(gs_heads->side, h->comm, h->comm_fork));
      DRV(allreduce(allreduce_ctx, 4));
      DRV_HIP(ev_record(h->comm_done, h->comm));
      return SERL_OK;
    };
    if (dp_overlap && !h->comm) {
      void* unused = nullptr;
      DRV(serl_learner_comm_stream(h, &unused));
    }
    DRV(features_bwd(h, 'C', critic, co.e, cgr, f, lo, B, ws, st, dinp, ldz, true, &h->gside, nslab, ss, &step_keys[8 * i + 6],
                     dp_overlap ? &bucket4 : nullptr, gs_heads));
    DRV(h->gside.join());
    DRV(h->gside2.join());
    DRV(phase(h, "features_bwd", st));
    if (dp_overlap) {
      DRV(allreduce(allreduce_ctx, 5));
      DRV_HIP(ev_wait(st, h->comm_done));
    } else if (world > 1) {
      DRV(allreduce(allreduce_ctx, 0));   // one all-reduce(sum) of the flat gradient per step
    }
    ++critic_step;
    float bc1, bc2;
    bias_corrections(critic_step, &bc1, &bc2);
    DRV(serl_adam_polyak(critic.base, ag->critic_mu, ag->critic_nu, cgr.base, target.base, co.size, c.critic_lr, ADAM_B1,
                         ADAM_B2, ADAM_EPS, bc1, bc2, c.tau, 1.0f, st));
  }

  // ---- actor step on the last minibatch (sac_learner.py:120-144, 235)
  DRV(phase(h, "actor", st));
  const int lo = (utd - 1) * B;
  // the actor's own pass over obs ran ahead; only the critic pass with the post-UTD critic parameters is left here
  if (sta != st) DRV_HIP(ev_wait(st, h->aobs));
  {
    FPass p{'C', critic, &co.e, 0, zP, st, key_drop_pi};
    DRV(features(h, &p, 1, f, lo, B, ws, st, 0, nullptr, true));
  }
  float* q;
  DRV(ensemble_fwd(h, 'C', critic, co, 0, 1, E, zP, ldz, B, ws, st, true, &q, key_drop_pi, rank * B, Bg));
  DRV(serl_actor_loss(q, E, B, logpA, log_temp, info + 2, st));
  DRV(serl_fill(dq, -1.0f / ((float)E * Bg), (long long)E * B, st));
  float* dinp = nullptr; int nslab = 1; long long ss = 0;
  DRV(ensemble_bwd(h, 'C', critic, co, nullptr, dq, zP, ldz, B, ws, st, nullptr, 1, &dinp, &nslab, &ss, key_drop_pi, rank * B, Bg));
  WSF(ws, st, dms, (size_t)B * 2 * A, "dms");
  WSF(ws, st, msA, (size_t)B * 2 * A, "ms_A");
  WSF(ws, st, epsA, (size_t)B * A, "eps_A");
  DRV(serl_tanh_gaussian_bwd(zP + c.latent, ldz, epsA, msA + A, 2 * A, dinp + c.latent, ldz, log_temp, 1.0f / Bg, dms,
                             dms + A, 2 * A, B, A, st));
  float* dzlat = nullptr;
  if (c.pixel) DRV(policy_bwd(h, 'A', actor, ao, agr, zA, ldz, B, ws, st, true, &dzlat));
  else DRV(policy_bwd(h, 'A', actor, ao, agr, f.obs + (size_t)lo * c.obs_dim, c.obs_dim, B, ws, st, false, &dzlat));
  DRV(features_bwd(h, 'A', actor, ao.e, agr, f, lo, B, ws, st, dzlat, c.latent, false, nullptr, 1, 0));
  if (world > 1) {
    DRV(allreduce(allreduce_ctx, 1));   // actor gradient
    DRV(allreduce(allreduce_ctx, 2));   // loss / q / logp sums -> global before the temperature step
  }
  float bc1, bc2;
  bias_corrections(ag->actor_step + 1, &bc1, &bc2);
  DRV(serl_adam_polyak(actor.base, ag->actor_mu, ag->actor_nu, agr.base, nullptr, ao.size, c.actor_lr, ADAM_B1, ADAM_B2,
                       ADAM_EPS, bc1, bc2, 0.0f, 1.0f, st));
  // ---- temperature (sac_learner.py:146-158, 236)
  if (c.has_temp) {
    bias_corrections(ag->temp_step + 1, &bc1, &bc2);
    DRV(serl_temperature_update(log_temp, info + 3, 1.0f / Bg, c.target_entropy, c.temp_lr, ADAM_B1, ADAM_B2, ADAM_EPS, bc1,
                                bc2, info + 4, st));
  }
  DRV(phase(h, "end", st));
  return SERL_OK;
}

// One policy evaluation on a single observation (serl/agents/agent.py:12-37): encoder + trunk + TanhNormal head with the
// given actor parameters.  act_out (A floats) / eps_out (A floats, may be NULL) are device pointers.
extern "C" int serl_learner_act(serl_learner* h, const float* actor_params, const unsigned char* const* images,
                                const float* state_or_obs, const unsigned* key, int deterministic, float* act_out,
                                float* eps_out, void* stream) {
  SERL_CHECK_ARG(h && actor_params && state_or_obs && act_out, "learner_act: null pointer");
  std::lock_guard<std::mutex> lock(h->act_mu);
  const Cfg& c = h->cfg;
  hipStream_t st = (hipStream_t)stream;
  Workspace& ws = h->ws_act;
  const Store actor{(float*)actor_params};
  WSF(ws, st, z, (size_t)c.ldz, "zA");
  if (c.pixel && c.encoder == 1) {
    SERL_CHECK_ARG(images && h->backbone, "learner_act: images or backbone missing");
    const long long wsb = serl_mbv3_workspace_bytes(1, c.H, c.W);
    WSF(ws, st, bbws, (size_t)wsb / 4, "bb_ws");
    Feats f;
    f.row0 = 0; f.total_rows = 1;
    for (int ci = 0; ci < c.ncam; ++ci) {
      WSF(ws, st, bb, (size_t)c.fh * c.fw * c.fc, "bb_0_%d", ci);
      DRV(serl_mbv3_forward(h->backbone, images[ci], 1, 1, c.H, c.W, bbws, wsb, bb, st));
      f.bb[0][ci] = bb;
    }
    FPass p{'A', actor, &h->ao.e, 0, z, st};       // training=False: no Dropout (agent.py:12-22)
    DRV(mobilenet_sle_fwd(h, &p, 1, f, 0, 1, ws, st));
    DRV(trunk_fwd(h, 'A', actor, h->ao.e, state_or_obs, z, c.ldz, 1, ws, st, false));
  } else if (c.pixel) {
    SERL_CHECK_ARG(images, "learner_act: images missing");
    Pass p; p.tag = 'A'; p.store = actor; p.eo = &h->ao.e;
    for (int ci = 0; ci < c.ncam; ++ci) p.imgs[ci] = images[ci];
    DRV(encoder_fwd(h, &p, 1, 1, ws, st, 0, true));
    DRV(trunk_fwd(h, 'A', actor, h->ao.e, state_or_obs, z, c.ldz, 1, ws, st, false));
  } else {
    DRV(serl_copy2d(z, c.ldz, state_or_obs, c.obs_dim, 1, c.obs_dim, st));
  }
  const bool save = eps_out != nullptr;
  DRV(policy_fwd(h, 'A', actor, h->ao, z, c.ldz, 1, ws, st, key, 0, 1, act_out, c.A, deterministic != 0, save, false, nullptr,
                 nullptr));
  if (save) {
    WSF(ws, st, eps, (size_t)c.A, "eps_A");
    DRV_HIP(hipMemcpyAsync(eps_out, eps, c.A * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  return SERL_OK;
}

// ================================================================= behaviour cloning
// One `update` of PixelBCLearner / PixelHybridBCLearner (serl/agents/bc/pixel_bc_learner.py:104-139,
// pixel_hybrid_bc_learner.py:135-185): random-shift augmentation of the observations with key_aug, then
//   actor:          nll = -mean log_prob(clip(a)) of the tanh-Gaussian policy, gradients THROUGH the image encoder
//                   (PixelMultiplexer(stop_gradient=False)), dropout key key1, one Adam step;
//   discrete actor: (hybrid, bc->disc != NULL) PixelMultiplexer(stop_gradient=True) + MLP + Dense(1) on the same augmented
//                   observations, sigmoid BCE against (last action column > 0), dropout key key2, one Adam step --
//                   the handle's "critic" slot describes this network (one member, no LayerNorm, no action input).
// The continuous policy sees the first action_dim columns of the (n, ld_actions) action rows.  info_out[0] = sum of the
// per-row nll, info_out[1] = sum of the per-row BCE.  Single rank; everything is enqueued on `stream`.
extern "C" int serl_learner_bc_update(serl_learner* h, const serl_bc_buffers* bc, const serl_update_batch* bt,
                                      int ld_actions, const unsigned* key_aug, const unsigned* key1, const unsigned* key2,
                                      float* info_out, void* stream) {
  SERL_CHECK_ARG(h && bc && bt && key_aug && key1 && key2 && info_out, "learner_bc_update: null pointer");
  const Cfg& c = h->cfg;
  SERL_CHECK_ARG(c.pixel, "learner_bc_update: the BC learners are pixel learners");
  SERL_CHECK_ARG(bc->actor && bc->actor_mu && bc->actor_nu && bc->actor_grads, "learner_bc_update: actor buffers missing");
  SERL_CHECK_ARG(bt->n >= 1 && bt->actions && bt->state, "learner_bc_update: actions / state missing");
  const bool hybrid = bc->disc != nullptr;
  SERL_CHECK_ARG(!hybrid || (bc->disc_mu && bc->disc_nu && bc->disc_grads && c.E == 1 && !c.ln && c.crit_A == 0),
                 "learner_bc_update: the discrete actor needs its buffers and a one-member, no-LayerNorm, no-action head");
  SERL_CHECK_ARG(ld_actions >= c.A + (hybrid ? 1 : 0), "learner_bc_update: action rows of %d columns", ld_actions);
  const int n = bt->n, A = c.A, ldz = c.ldz;
  hipStream_t st = (hipStream_t)stream;
  Workspace& ws = h->ws_update[n];
  const Store actor{bc->actor}, agr{bc->actor_grads};
  const uint32_t ka[2] = {key_aug[0], key_aug[1]}, k1[2] = {key1[0], key1[1]}, k2[2] = {key2[0], key2[1]};

  Feats f;
  f.row0 = 0; f.total_rows = n;
  f.state = bt->state; f.next_state = bt->state;
  DRV(augment_side(h, bt->pixels, 0, ka, n, 1, n, 0, 1, ws, st, f));
  if (c.encoder == 1) DRV(backbone_side(h, 0, n, ws, st, f));

  WSF(ws, st, zA, (size_t)n * ldz, "zA");
  WSF(ws, st, dms, (size_t)n * 2 * A, "dms");
  {
    FPass p{'A', actor, &h->ao.e, 0, zA, st, k1};
    DRV(features(h, &p, 1, f, 0, n, ws, st, 'A', nullptr, true));
  }
  DRV(policy_fwd(h, 'A', actor, h->ao, zA, ldz, n, ws, st, nullptr, 0, n, nullptr, 0, true, false, false, nullptr, nullptr));
  WSF(ws, st, msA, (size_t)n * 2 * A, "ms_A");
  DRV(serl_tanh_gaussian_nll(msA, msA + A, 2 * A, bt->actions, ld_actions, n, A, 1.0f / (float)n, dms, dms + A, 2 * A,
                             info_out, st));
  float* dzlat = nullptr;
  DRV(policy_bwd(h, 'A', actor, h->ao, agr, zA, ldz, n, ws, st, true, &dzlat));
  DRV(features_bwd(h, 'A', actor, h->ao.e, agr, f, 0, n, ws, st, dzlat, c.latent, true, nullptr, 1, 0, k1));
  float bc1, bc2;
  bias_corrections(bc->actor_step + 1, &bc1, &bc2);
  DRV(serl_adam_polyak(actor.base, bc->actor_mu, bc->actor_nu, agr.base, nullptr, h->ao.size, c.actor_lr, ADAM_B1, ADAM_B2,
                       ADAM_EPS, bc1, bc2, 0.0f, 1.0f, st));
  if (!hybrid) return SERL_OK;

  const Store disc{bc->disc}, dgr{bc->disc_grads};
  WSF(ws, st, zC, (size_t)n * ldz, "zC");
  WSF(ws, st, dq, (size_t)n, "dq");
  {
    FPass p{'C', disc, &h->co.e, 0, zC, st, k2};
    DRV(features(h, &p, 1, f, 0, n, ws, st, 'C', nullptr, true));
  }
  float* logits = nullptr;
  DRV(ensemble_fwd(h, 'C', disc, h->co, 0, 1, 1, zC, ldz, n, ws, st, true, &logits));
  DRV(serl_bce_logits(logits, bt->actions + A, ld_actions, n, 1.0f / (float)n, dq, info_out + 1, st));
  float* dinp = nullptr; int nslab = 1; long long ss = 0;
  DRV(ensemble_bwd(h, 'C', disc, h->co, &dgr, dq, zC, ldz, n, ws, st, nullptr, 1, &dinp, &nslab, &ss));
  // stop_gradient behind the encoder: trunk gradients only; the encoder leaves of disc_grads stay zero (never written)
  DRV(features_bwd(h, 'C', disc, h->co.e, dgr, f, 0, n, ws, st, dinp, c.latent, false, nullptr, 1, 0, k2));
  bias_corrections(bc->disc_step + 1, &bc1, &bc2);
  return serl_adam_polyak(disc.base, bc->disc_mu, bc->disc_nu, dgr.base, nullptr, h->co.size, c.actor_lr, ADAM_B1, ADAM_B2,
                          ADAM_EPS, bc1, bc2, 0.0f, 1.0f, st);
}

// The discrete actor's logit for ONE observation (PixelHybridBCLearner.compute_actions, pixel_hybrid_bc_learner.py:126):
// encoder + trunk + MLP + Dense(1) with the given parameters, training=False.  logit_out: one device float.
extern "C" int serl_learner_bc_logit(serl_learner* h, const float* disc_params, const unsigned char* const* images,
                                     const float* state, float* logit_out, void* stream) {
  SERL_CHECK_ARG(h && disc_params && images && state && logit_out, "learner_bc_logit: null pointer");
  std::lock_guard<std::mutex> lock(h->act_mu);
  const Cfg& c = h->cfg;
  SERL_CHECK_ARG(c.pixel && c.E == 1 && !c.ln && c.crit_A == 0, "learner_bc_logit: not a hybrid BC learner");
  hipStream_t st = (hipStream_t)stream;
  Workspace& ws = h->ws_act;
  const Store disc{(float*)disc_params};
  WSF(ws, st, z, (size_t)c.ldz, "zC");
  if (c.encoder == 1) {
    SERL_CHECK_ARG(h->backbone, "learner_bc_logit: backbone missing");
    const long long wsb = serl_mbv3_workspace_bytes(1, c.H, c.W);
    WSF(ws, st, bbws, (size_t)wsb / 4, "bb_ws");
    Feats f;
    f.row0 = 0; f.total_rows = 1;
    for (int ci = 0; ci < c.ncam; ++ci) {
      WSF(ws, st, bb, (size_t)c.fh * c.fw * c.fc, "bb_0_%d", ci);
      DRV(serl_mbv3_forward(h->backbone, images[ci], 1, 1, c.H, c.W, bbws, wsb, bb, st));
      f.bb[0][ci] = bb;
    }
    FPass p{'C', disc, &h->co.e, 0, z, st};
    DRV(mobilenet_sle_fwd(h, &p, 1, f, 0, 1, ws, st));
  } else {
    Pass p; p.tag = 'C'; p.store = disc; p.eo = &h->co.e;
    for (int ci = 0; ci < c.ncam; ++ci) p.imgs[ci] = images[ci];
    DRV(encoder_fwd(h, &p, 1, 1, ws, st, 0, true));
  }
  DRV(trunk_fwd(h, 'C', disc, h->co.e, state, z, c.ldz, 1, ws, st, false));
  float* q = nullptr;
  DRV(ensemble_fwd(h, 'C', disc, h->co, 0, 1, 1, z, c.ldz, 1, ws, st, false, &q));
  DRV_HIP(hipMemcpyAsync(logit_out, q, sizeof(float), hipMemcpyDeviceToDevice, st));
  return SERL_OK;
}

// Copy `nbytes` of a named workspace buffer of the last update at minibatch size B into dst (device), on `stream`:
// tests read the augmented images ("aug_<side>_<cam>", side 0 obs / 1 next_obs, cam in sorted order) this way.
extern "C" int serl_learner_read_buffer(serl_learner* h, int B, const char* name, void* dst, long long nbytes, void* stream) {
  SERL_CHECK_ARG(h && name && dst && nbytes >= 0, "learner_read_buffer: bad args");
  auto w = h->ws_update.find(B);
  SERL_CHECK_ARG(w != h->ws_update.end(), "learner_read_buffer: no update with minibatch size %d has run", B);
  const Workspace::Buf* it = w->second.find(name);
  SERL_CHECK_ARG(it != nullptr, "learner_read_buffer: no buffer named %s", name);
  SERL_CHECK_ARG((size_t)nbytes <= it->bytes, "learner_read_buffer: %s holds %zu bytes", name, it->bytes);
  DRV_HIP(hipMemcpyAsync(dst, it->p, (size_t)nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return SERL_OK;
}

extern "C" int serl_learner_buffer_ptr(serl_learner* h, int B, const char* name, void** ptr, long long* nbytes) {
  SERL_CHECK_ARG(h && name && ptr, "learner_buffer_ptr: bad args");
  auto w = h->ws_update.find(B);
  SERL_CHECK_ARG(w != h->ws_update.end(), "learner_buffer_ptr: no update with minibatch size %d has run", B);
  const Workspace::Buf* it = w->second.find(name);
  SERL_CHECK_ARG(it != nullptr, "learner_buffer_ptr: no buffer named %s", name);
  *ptr = it->p;
  if (nbytes) *nbytes = (long long)it->bytes;
  return SERL_OK;
}

// The stream the data-parallel hook must enqueue bucket 4 on (which = 4); created on first use.
extern "C" int serl_learner_comm_stream(serl_learner* h, void** stream) {
  SERL_CHECK_ARG(h && stream, "learner_comm_stream: null pointer");
  if (!h->comm) {
    DRV_HIP(hipStreamCreateWithFlags(&h->comm, hipStreamNonBlocking));
    DRV_HIP(hipEventCreateWithFlags(&h->comm_fork, hipEventDisableTiming));
    DRV_HIP(hipEventCreateWithFlags(&h->comm_done, hipEventDisableTiming));
  }
  *stream = (void*)h->comm;
  return SERL_OK;
}

extern "C" long long serl_learner_workspace_bytes(serl_learner* h) {
  if (!h) return 0;
  long long t = 0;
  for (auto& kv : h->ws_update)
    for (auto& b : kv.second.m) t += (long long)b.second.bytes;
  for (auto& b : h->ws_act.m) t += (long long)b.second.bytes;
  return t;
}

// ---- instrumentation for bench.py: HIP events around every conv launch (on the launch stream) / between phases
extern "C" int serl_learner_set_timing(serl_learner* h, int conv_events, int phase_events) {
  SERL_CHECK_ARG(h, "learner_set_timing: null handle");
  h->timing = conv_events != 0;
  h->phases = phase_events != 0;
  return SERL_OK;
}
extern "C" int serl_learner_timing_count(serl_learner* h) { return h ? (int)h->trecs.size() : 0; }
// record i -> desc[7] = {kind (0 fwd, 1 dgrad, 2 wgrad, 3 fused layers 1+2 fwd, 4 conv2 dgrad + conv1 wgrad, 5 fused layers 1+2 fwd keeping layer 1 for half the groups), in_u8, groups, B, IH, C, F}, val[3] = {ms, flops, bytes};
// call after a device synchronise.
extern "C" int serl_learner_timing_get(serl_learner* h, int i, int* desc, double* val) {
  SERL_CHECK_ARG(h && desc && val && i >= 0 && i < (int)h->trecs.size(), "learner_timing_get: bad index");
  const TimeRec& r = h->trecs[i];
  float ms = 0;
  DRV_HIP(hipEventElapsedTime(&ms, r.a, r.b));
  desc[0] = r.kind; desc[1] = r.in_u8; desc[2] = r.groups; desc[3] = r.B; desc[4] = r.IH; desc[5] = r.C; desc[6] = r.F;
  val[0] = ms; val[1] = r.flops; val[2] = r.bytes;
  return SERL_OK;
}
extern "C" int serl_learner_phase_count(serl_learner* h) { return h ? (int)h->precs.size() : 0; }
// mark i: name (static string) and the milliseconds from mark i to mark i+1 (0 for the last one)
extern "C" int serl_learner_phase_get(serl_learner* h, int i, const char** name, double* ms_to_next) {
  SERL_CHECK_ARG(h && name && ms_to_next && i >= 0 && i < (int)h->precs.size(), "learner_phase_get: bad index");
  *name = h->precs[i].name;
  float ms = 0;
  if (i + 1 < (int)h->precs.size()) DRV_HIP(hipEventElapsedTime(&ms, h->precs[i].ev, h->precs[i + 1].ev));
  *ms_to_next = ms;
  return SERL_OK;
}
extern "C" int serl_learner_timing_reset(serl_learner* h) {
  SERL_CHECK_ARG(h, "learner_timing_reset: null handle");
  for (auto& r : h->trecs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (auto& r : h->precs) (void)hipEventDestroy(r.ev);
  h->trecs.clear();
  h->precs.clear();
  return SERL_OK;
}
