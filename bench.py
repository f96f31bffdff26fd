"""Headline benchmark: learner grad-steps/sec, DrQ-RLPD 128x128 pixels, batch 256, 10 critics, UTD 20
(BASELINE.json `metric`; SURVEY.md 8d config C3 with RLPD settings).

One "step" = one `agent.update(batch, utd_ratio=20)` call = 20 critic grad-steps + 1 actor/temperature
step + the DrQ gather/unpack/crop of the whole 5120-sample batch.  Inputs (replay frame store, sampled
indices, states, actions) are resident in HBM when the timed region starts.
`value` = grad-steps/s x (global_batch / 256): at N GPUs every rank keeps batch 256 (weak scaling, one RCCL
gradient all-reduce per critic step), so value = batch-256-equivalent grad-steps/s of the whole job.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--global-batch G] [--no-cpu-baseline]
N > 1: one rank per GPU.  Under torch.distributed.run (WORLD_SIZE set) this process is one of the ranks; started
plainly (`python bench.py --gpus N`) it starts the N ranks itself as children of `python -m torch.distributed.run`,
BEFORE anything touches the GPU, relays rank 0's JSON line and exits with the children's return code.
`--global-batch G` (BASELINE config 5: 4096 / 8192) gives every rank G/N rows per critic step (strong scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

KEYS = ("wrist_1", "wrist_2")
H = W = 128
STACK, STATE_DIM, ACT_DIM = 3, 19, 7
BATCH, UTD = 256, 20
AGENT_KW = dict(cnn_features=(32, 64, 128, 256), cnn_filters=(3, 3, 3, 3), cnn_strides=(2, 2, 2, 2), latent_dim=64,
                num_qs=10, num_min_qs=2, critic_layer_norm=True, backup_entropy=False, init_temperature=0.1,
                hidden_dims=(256, 256), discount=0.99, tau=0.005)
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (no sparsity)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FLOP_PER_GRAD_STEP = 304e9        # 301 critic + 3.1 amortised actor (2 FLOP / MAC)


def fill_buffer(buf, n_transitions, seed):
    from helpers import rollout_transitions
    rng = np.random.default_rng(seed)
    for tr in rollout_transitions(rng, n_transitions, KEYS, H, W, STACK, STATE_DIM, ACT_DIM, episode_len=100):
        buf.insert(tr)


def measured_traffic(key):
    """HBM-side bytes per launch of a conv launch configuration, from the newest profiles/*_traffic.json
    (tools/traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 correction applied).
    bench.py cannot collect PMC counters on itself; None when no measurement of this configuration is on file."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None, None
    kind, in_u8, groups, B, IH, C, F = key
    layer = {H: 1, (H - 3) // 2 + 1: 2, ((H - 3) // 2 + 1 - 3) // 2 + 1: 3}.get(IH, 4)
    op = f"conv{layer} {kind.split('_')[-1]} g{groups}"
    if B != BATCH or not kind.startswith("convs_"):
        return None, None
    e = json.load(open(files[-1]))["launches"].get(op)
    return (e["hbm_side_bytes"], os.path.basename(files[-1])) if e else (None, None)


def cpu_baseline(seed=0, progress=lambda m: None):
    """CPU restatement (oracle, torch-CPU fp32 -- NOT JAX/XLA) on a bounded sample of the same workload, BASELINE.md
    section 3's protocol cut to the bench's time budget: update() calls at UTD 20, batch 256 (20 grad-steps + the
    actor/temperature step each), one un-timed warm-up call at UTD 2 (thread pool, allocator), then the median of 3
    timed calls (the protocol asks for >= 5 after 2 warm-ups: ~15 s per call on 16 threads makes that > 100 s)."""
    import torch
    from helpers import agent_to_oracle, make_pixel_batch, pixel_spaces
    from oracle import learner as ol
    from serl_amd.agents import DrQLearner
    obs_space, act_space = pixel_spaces(KEYS, H, W, STACK, STATE_DIM, ACT_DIM)
    agent = DrQLearner.create(seed, obs_space, act_space, pixel_keys=KEYS, **AGENT_KW)
    ostate, ocfg = agent_to_oracle(agent)
    u, ncalls = 20, 3
    rng = np.random.default_rng(seed)
    warm = make_pixel_batch(rng, 256 * 2, KEYS, H, W, STACK, STATE_DIM, ACT_DIM, packed=True)
    ostate, _ = ol.drq_update(ostate, ocfg, warm, 2)
    batch = make_pixel_batch(rng, 256 * u, KEYS, H, W, STACK, STATE_DIM, ACT_DIM, packed=True)
    times = []
    for _ in range(ncalls):
        t0 = time.perf_counter()
        ostate, _ = ol.drq_update(ostate, ocfg, batch, u)
        times.append(time.perf_counter() - t0)
        progress(f"  cpu call {len(times)}/{ncalls}: {times[-1]:.1f} s")
    dt = float(np.median(times))
    return {"value": u / dt, "unit": "grad-steps/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"median of {ncalls} update() calls at UTD {u}, batch 256 ({u} grad-steps + actor step each; "
                      f"{', '.join(f'{t:.1f}' for t in times)} s) after one warm-up call at UTD 2; "
                      "CPU restatement (torch-CPU fp32), not JAX/XLA"}


def _timed_calls(torch, fn, n, warm=2):
    """n calls of fn() between synchronisations after `warm` un-timed ones -> seconds per call."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def extra_configs(progress):
    """The other configurations north_star states its target on (BASELINE.json configs[1], [3], [4]), driver-visible as
    extra keys of the JSON line; the headline `value` is untouched.  All at the library's default precision (bf16_3x),
    synthetic data resident in HBM, same definition of a grad-step as the headline."""
    import torch
    from helpers import make_pixel_batch, make_state_batch, pixel_spaces, state_spaces
    from serl_amd.agents import DrQLearner, SACLearner, VICELearner
    from serl_amd.data import MemoryEfficientReplayBuffer
    from serl_amd.data.replay_buffer import to_device
    dev = torch.device("cuda")
    out = {}
    # ---- C2: state-based RLPD (obs 17, act 6, 10/2 LayerNorm critics, no entropy backup), batch 256, UTD 20
    obs_space, act_space = state_spaces(17, 6)
    agent = SACLearner.create(42, obs_space, act_space, num_qs=10, num_min_qs=2, critic_layer_norm=True,
                              backup_entropy=False, init_temperature=0.1)
    rng = np.random.default_rng(0)
    sb = [to_device(make_state_batch(rng, 256 * 20, 17, 6), dev) for _ in range(4)]
    st = {"a": agent, "i": 0}

    def c2_call():
        st["a"], _ = st["a"].update(sb[st["i"] % 4], 20)
        st["i"] += 1
    t = _timed_calls(torch, c2_call, 20, warm=3)
    out["c2_us_per_grad_step"] = round(t / 20 * 1e6, 1)
    out["c2_value"] = round(20 / t, 1)
    out["c2_note"] = ("BASELINE config 2: state RLPD update(), batch 256, UTD 20, 10/2 LayerNorm critics; launch-latency "
                      "bound (1.2 GFLOP, 26 MB per grad-step): us per grad-step is the figure of merit; 20 timed calls")
    progress(f"c2: {out['c2_us_per_grad_step']} us per grad-step")
    del agent, st, sb
    # ---- C5 shape on one GPU: the headline learner at batch 4096, UTD 1
    obs_space, act_space = pixel_spaces(KEYS, H, W, STACK, STATE_DIM, ACT_DIM)
    agent = DrQLearner.create(42, obs_space, act_space, pixel_keys=KEYS, **AGENT_KW)
    buf = MemoryEfficientReplayBuffer(obs_space, act_space, 3000, pixel_keys=KEYS, device="cuda")
    fill_buffer(buf, 2000, seed=1)
    buf.seed(1)
    it = buf.get_iterator(queue_size=1, sample_args=dict(batch_size=4096, pack_obs_and_next_obs=True,
                                                         demo_batch_size=2048, demo_size=500))
    pb = [next(it) for _ in range(3)]
    st = {"a": agent, "i": 0}

    def c5_call():
        st["a"], _ = st["a"].update(pb[st["i"] % 3], 1, pixel_keys=KEYS)
        st["i"] += 1
    t = _timed_calls(torch, c5_call, 6, warm=2)
    out["c5_1gpu_value"] = round(1 / t * 4096 / 256, 1)
    out["c5_1gpu_ms_per_update"] = round(t * 1e3, 2)
    out["c5_note"] = ("BASELINE config 5 per-GPU shape on ONE GPU: DrQ-RLPD update(), batch 4096, UTD 1; value in 256-row "
                      "grad-step equivalents per second (every update carries its own actor + temperature step); 6 timed calls")
    progress(f"c5 (1 GPU, batch 4096, UTD 1): {out['c5_1gpu_value']} grad-step equivalents/s")
    del agent, st, pb, it, buf
    torch.cuda.empty_cache()
    # ---- C4: frozen MobileNetV3 encoder + VICE rewards inside update(), stack 1, batch 256, UTD 20, and one
    # update_classifier per 100 update() calls (binpick_franka_fwbw_vice.py:59,86-89)
    obs_space, act_space = pixel_spaces(KEYS, H, W, 1, STATE_DIM, ACT_DIM)
    kw = {k: v for k, v in AGENT_KW.items() if not k.startswith("cnn_")}
    pool = {k: rng.integers(0, 256, size=(64, H, W, 3, 1), dtype=np.uint8) for k in KEYS}
    agent = VICELearner.create(42, obs_space, act_space, pixel_keys=KEYS, encoder="mobilenet", vice_dropout_rate=0.1,
                               vice_goal_pool=pool, **kw)
    vb = [to_device(make_pixel_batch(rng, 256 * 20, KEYS, H, W, 1, STATE_DIM, ACT_DIM, packed=True), dev) for _ in range(2)]
    cb = to_device(make_pixel_batch(rng, 256, KEYS, H, W, 1, STATE_DIM, ACT_DIM, packed=True), dev)
    st = {"a": agent, "i": 0}

    def c4_call():
        st["a"], _ = st["a"].update(vb[st["i"] % 2], 20, pixel_keys=KEYS)
        st["i"] += 1

    def c4_clf():
        st["a"], _ = st["a"].update_classifier(cb, KEYS)
    t_upd = _timed_calls(torch, c4_call, 4, warm=2)
    t_clf = _timed_calls(torch, c4_clf, 3, warm=1)
    t = t_upd + t_clf / 100.0
    out["c4_value"] = round(20 / t, 1)
    out["c4_ms_per_update"] = round(t_upd * 1e3, 2)
    out["c4_update_classifier_ms"] = round(t_clf * 1e3, 2)
    # the frozen backbone is the dominant family of this config: its forward over the 4 x 5120 augmented images of a call
    bb = agent._shared["backbone"]
    NB = 1024                                             # images per launch chain, as the step driver chunks a call's 5120
    imgs = torch.from_numpy(rng.integers(0, 256, size=(NB, H, W, 3), dtype=np.uint8)).to(dev)
    fo = torch.empty((NB,) + tuple(bb.feature_shape(H, W)), dtype=torch.float32, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    bb.forward(imgs, out=fo)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        bb.forward(imgs, out=fo)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    flops = 2 * 72e6 * (H * W) / (128 * 128) * NB         # 72 MMAC per 128 x 128 image
    byts = 7.0e6 * (H * W) / (128 * 128) * NB             # ~7 MB of fp32 NHWC activation traffic per image
    ach = byts / (ms * 1e-3) / 1e9
    out["c4_roofline"] = {"kernel": "frozen MobileNetV3-Large forward (stem + 15 blocks: depthwise / SE kernels + streaming / igemm "
                                    f"pointwise convs), {NB} images per launch chain", "bound": "hbm", "achieved": round(ach, 1),
                          "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4),
                          "avg_chain_ms": round(ms, 3), "algorithmic_bytes": byts, "algorithmic_flops": flops,
                          "tflops": round(flops / (ms * 1e-3) / 1e12, 2), "traffic": None,
                          "share_of_update": round(ms * (4 * 20 * 256 / NB) / (t_upd * 1e3), 3),
                          "timing": f"torch events on the launch stream around 3 repeats of the {NB}-image chain, stand-alone"}
    out["c4_note"] = ("BASELINE config 4: DrQ-RLPD with encoder='mobilenet' (frozen synthetic MobileNetV3-Large) + VICE rewards "
                      "in update(), 2 cams 128x128x3, stack 1, batch 256, UTD 20; value = 20 / (update + update_classifier/100) "
                      "grad-steps/s; 4 timed update() calls, 3 update_classifier calls")
    progress(f"c4: {out['c4_value']} grad-steps/s ({out['c4_ms_per_update']} ms per update, classifier step "
             f"{out['c4_update_classifier_ms']} ms)")
    return out


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (torch.distributed.run, rendezvous
    on 127.0.0.1) before this process imports torch or touches the GPU; relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line)
    elif proc.returncode == 0:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        return 1
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU minibatch (default 256 = the headline config; "
                    "other values are SURVEY 8d's large-batch variants and are labelled as such)")
    ap.add_argument("--utd", type=int, default=None, help="UTD ratio (default 20 = the headline config)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="rows per critic step over ALL ranks (BASELINE config 5: 4096 / 8192): every rank takes "
                         "global_batch / gpus rows (strong scaling); default: 256 rows per rank (weak scaling)")
    ap.add_argument("--no-highest", action="store_true", help="skip the extra timed steps at precision 'highest'")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the c2 / c4 / c5 legs (state RLPD, MobileNet + VICE, batch 4096 on one GPU)")
    ap.add_argument("--phases", action="store_true",
                    help="after the timed region, one more update() with HIP-event marks between its phases (stderr)")
    ap.add_argument("--precision", default="bf16_3x", choices=["bf16_3x", "highest"],
                    help="serl_amd.set_matmul_precision: bf16_3x (default of the library) or highest (exact fp32 MFMA)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))       # nothing has touched the GPU (torch is not even imported)
    global BATCH, UTD
    if args.global_batch is not None:
        assert args.batch is None, "--batch (rows per rank) and --global-batch (rows over all ranks) exclude each other"
        assert args.global_batch % args.gpus == 0, "--global-batch must be divisible by --gpus"
        args.batch = args.global_batch // args.gpus
    headline = (args.batch in (None, 256)) and (args.utd in (None, 20)) and args.global_batch is None
    BATCH, UTD = args.batch or BATCH, args.utd or UTD

    import torch
    from helpers import pixel_spaces
    from serl_amd import ops
    from serl_amd.agents import DrQLearner
    from serl_amd.data import MemoryEfficientReplayBuffer
    import serl_amd
    serl_amd.set_matmul_precision(args.precision)
    precision = serl_amd.get_matmul_precision()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal of the N > 1 path on a one-GPU box: SERL_AMD_BENCH_REHEARSAL=1 -> every rank on cuda:0, gloo backend
    rehearsal = os.environ.get("SERL_AMD_BENCH_REHEARSAL", "0") != "0"
    if rehearsal:
        local_rank = 0
        if os.environ.get("REH_DUMP_AFTER"):     # a stalled rank prints the stacks of all its threads and exits
            import faulthandler
            faulthandler.dump_traceback_later(int(os.environ["REH_DUMP_AFTER"]), exit=True)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as td
        from serl_amd.dist import DataParallel
        if rehearsal:
            td.init_process_group("gloo")
        else:
            td.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist = DataParallel()

    t_start = time.perf_counter()

    def progress(msg):          # stderr, rank 0: a line per phase so that a long run never looks hung
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    obs_space, act_space = pixel_spaces(KEYS, H, W, STACK, STATE_DIM, ACT_DIM)
    agent = DrQLearner.create(42, obs_space, act_space, pixel_keys=KEYS, dist=dist, **AGENT_KW)
    progress(f"agent created (world {world}, batch {BATCH}/gpu, utd {UTD}, precision {precision})")
    # online buffer + demo prefix: 50/50 demo/online mixing via the tail-overwrite rule
    buf = MemoryEfficientReplayBuffer(obs_space, act_space, 6000, pixel_keys=KEYS, device="cuda")
    fill_buffer(buf, 4000, seed=rank)
    buf.seed(rank)
    sample_args = dict(batch_size=BATCH * UTD, pack_obs_and_next_obs=True, demo_batch_size=BATCH * UTD // 2,
                       demo_size=1000)
    it = buf.get_iterator(queue_size=1, sample_args=sample_args)
    t0 = time.perf_counter()
    batches = [next(it) for _ in range(args.warmup + args.steps)]
    sample_ms = (time.perf_counter() - t0) * 1e3 / len(batches)

    def barrier():
        if dist is not None:
            import torch.distributed as td
            td.barrier()
        torch.cuda.synchronize()

    progress("buffer filled, batches sampled")
    for b in batches[:args.warmup]:
        agent, info = agent.update(b, UTD, pixel_keys=KEYS)
    torch.cuda.synchronize()
    progress("warm-up done")
    native = agent._native
    if dist is not None:
        dist.timing_reset()
        dist.timing = True
    barrier()
    c0 = time.thread_time()
    t0 = time.perf_counter()
    for b in batches[args.warmup:]:
        agent, info = agent.update(b, UTD, pixel_keys=KEYS)
    host_enqueue = time.perf_counter() - t0           # update() does not wait for the GPU: this is pure launch time ...
    host_cpu = time.thread_time() - c0                # ... of which this much is CPU time of the calling thread (the rest:
    barrier()                                         # waiting on a full hardware queue)
    elapsed = time.perf_counter() - t0
    allreduce_s = None
    if dist is not None:
        dist.timing = False
        allreduce_s = dist.allreduce_seconds
    # roofline events: a second, short loop with HIP events around every conv launch (recorded by the driver on the
    # launch stream) -- kept out of the timed region above (1 640 extra hipEventRecord calls per step)
    timer = None
    if not args.no_kernel_timer:
        timer = native
        native.timing_reset()
        native.set_timing(conv_events=True)
        ragent = agent
        for b in batches[-min(2, len(batches)):]:
            ragent, _ = ragent.update(b, UTD, pixel_keys=KEYS)
        torch.cuda.synchronize()
        native.set_timing(conv_events=False)
        del ragent
    progress(f"timed region done: {elapsed / args.steps * 1e3:.2f} ms/step")
    # end to end: the same loop with the host-side index draw, rejection resampling and H2D of the small leaves inside
    # it (the iterator samples batch i+1 while the GPU runs update i: update() does not block the host)
    e2e = float("nan")
    if not rehearsal:       # (the one-GPU gloo rehearsal needs seconds per all-reduce: it runs the timed region only)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            agent, info = agent.update(next(it), UTD, pixel_keys=KEYS)
        barrier()
        e2e = time.perf_counter() - t0
    conv_records = native.conv_timings() if timer is not None else []
    if args.phases and rank == 0:
        native.timing_reset()
        native.set_timing(phase_events=True)
        agent.update(batches[-1], UTD, pixel_keys=KEYS)
        torch.cuda.synchronize()
        native.set_timing()
        marks = native.phase_timings()
        acc = {}
        for (n0, ms), (n1, _) in zip(marks[:-1], marks[1:]):
            acc.setdefault(f"{n0}->{n1}", []).append(ms)
        for k, v in acc.items():
            print(f"phase {k:28s} n={len(v):3d} mean {sum(v) / len(v) * 1e3:8.1f} us  total {sum(v):7.2f} ms",
                  file=sys.stderr)
    def max_over_ranks(vals):
        if dist is None:
            return vals
        import torch.distributed as td
        # gloo (one-GPU rehearsal): a host tensor -- torch's staging of device tensors for gloo has stalled here (dist.py)
        t = torch.tensor(vals, device="cpu" if rehearsal else "cuda", dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]
    elapsed, e2e = max_over_ranks([elapsed, e2e])

    # the same steps at precision "highest" (exact fp32 MFMA: the precision at which north_star's free-running
    # rel-1e-3 tolerance holds), timed by this run so that the figure is not only the builder's
    progress("end-to-end loop done")
    highest_value = None
    if precision != "highest" and not args.no_highest and not rehearsal:
        serl_amd.set_matmul_precision("highest")
        hb = batches[-min(3, len(batches)):]
        agent_h, _ = agent.update(hb[0], UTD, pixel_keys=KEYS)        # warm-up: fp32 workspaces
        barrier()
        t0 = time.perf_counter()
        for b in hb:
            agent_h, _ = agent_h.update(b, UTD, pixel_keys=KEYS)
        barrier()
        th = time.perf_counter() - t0
        th = max_over_ranks([th])[0]
        highest_value = len(hb) * UTD * world / th * (BATCH / 256.0)
        serl_amd.set_matmul_precision(precision)
        progress("precision 'highest' steps done")

    grad_steps = args.steps * UTD
    value = grad_steps * world / elapsed * (BATCH / 256.0)
    roofline = None
    if timer is not None:
        # dominant kernel = the conv launch configuration with the largest total time in the timed region; its
        # roof is the larger of the two lower bounds: algorithmic HBM bytes / 8 TB/s, and MFMA work
        # / dense peak (bf16_3x: 3 bf16 MFMA flops per algorithmic flop at 2500 TFLOP/s; highest: fp32 MFMA).
        # A kernel (one template instantiation, as rocprofv3 lists it) is launched with different group counts: the
        # target + critic passes (4 groups) on the main stream, the actor's run-ahead pass (2 groups) on its own.
        # Dominant kernel = largest total time over all its launches; figures are per launch, averaged over them.
        summ = {}
        for key, ms_, fl_, by_ in conv_records:
            if precision == "highest":
                key = (key[0].replace("convs_", "conv2d_"),) + key[1:]
            e = summ.setdefault(key, [0, 0.0, fl_, by_])
            e[0] += 1; e[1] += ms_
        summ = {k: (v[0], v[1] / v[0], v[2], v[3]) for k, v in summ.items()}
        fams = {}
        for k, (n_, ms_, fl_, by_) in summ.items():
            # (the fused layers-1+2 kernel is one instantiation whether or not some groups keep layer 1's output)
            f = fams.setdefault((k[0].replace("fwd12keep", "fwd12"),) + k[1:2] + k[3:],
                                dict(n=0, t=0.0, fl=0.0, by=0.0, tr=0.0, tr_ok=True, src=None, groups=[]))
            tr, src = measured_traffic(k)
            f["n"] += n_; f["t"] += n_ * ms_; f["fl"] += n_ * fl_; f["by"] += n_ * by_
            f["tr_ok"] = f["tr_ok"] and tr is not None
            f["tr"] += n_ * (tr or 0.0); f["src"] = src or f["src"]; f["groups"].append(k[2])
        fkey = max(fams, key=lambda k: fams[k]["t"])
        fam = fams[fkey]
        bf3 = precision == "bf16_3x"
        mfma_peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if bf3 else PEAK_FP32_MFMA_TFLOPS

        def roof(v):
            n, ms, flops, byts = v
            t_mfma, t_hbm = flops / (mfma_peak * 1e12), byts / (PEAK_HBM_GBS * 1e9)
            if t_hbm >= t_mfma:
                ach = byts / (ms * 1e-3) / 1e9
                return {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(ach / PEAK_HBM_GBS, 4)}
            ach = flops / (ms * 1e-3) / 1e12
            return {"bound": "mfma", "achieved": round(ach, 2), "peak": round(mfma_peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach / mfma_peak, 4)}
        n = fam["n"]
        ms, flops, byts = fam["t"] / n, fam["fl"] / n, fam["by"] / n
        roofline = roof((n, ms, flops, byts))
        roofline.update({
            "kernel": f"{fkey[0]} (u8_in={fkey[1]}, B={fkey[2]}, IH={fkey[3]}, C={fkey[4]}, F={fkey[5]}; launches with "
                      f"{sorted(fam['groups'])} groups, per-launch averages)",
            "traffic": fam["tr"] / n if fam["tr_ok"] else None, "traffic_source": fam["src"] if fam["tr_ok"] else None,
            "launches_timed": n, "avg_launch_ms": round(ms, 4),
            "concurrency_note": "durations are measured while other streams run kernels (weight gradient next to "
                                "data gradient, actor run-ahead pass next to the critic step): a kernel's share of "
                                "the HBM roof, not its stand-alone speed (tools/bench_conv.py)",
            "algorithmic_bytes_per_launch": byts, "algorithmic_flops_per_launch": flops,
            "mfma_peak_note": "2500 TFLOP/s dense bf16 / 3 MFMAs per product block" if bf3 else "fp32 MFMA",
            "all_conv_kernels": {f"{k[0]}|u8={k[1]}|g={k[2]}|IH={k[4]}|C={k[5]}|F={k[6]}":
                                 dict(roof(v), launches=v[0], avg_ms=round(v[1], 4),
                                      tflops=round(v[2] / (v[1] * 1e-3) / 1e12, 2),
                                      gbs=round(v[3] / (v[1] * 1e-3) / 1e9, 1)) for k, v in summ.items()}})
    gb = BATCH * world
    out = {
        "metric": "learner grad-steps/sec, DrQ-RLPD 128^2 pixels, batch 256, 10 critics, UTD 20" if headline else
                  f"grad-steps/sec x global_batch/256 (BASELINE config 5 family, NOT the headline config): DrQ-RLPD "
                  f"128^2 pixels, global batch {gb}, 10 critics, UTD {UTD}",
        "value": round(value, 3), "unit": "grad-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "strong" if args.global_batch is not None else "weak",
        "vs_baseline": None, "dtype": "f32" if precision == "highest" else "bf16_3x", "data": "synthetic",
        "config": {"workload": f"DrQ-RLPD update(): 2 cams 128x128x3 uint8, frame stack 3, conv encoder 32-64-128-256 "
                               f"stride 2, 10 LayerNorm critics / 2 subsampled targets, UTD {UTD}, batch {BATCH} per "
                               f"GPU, 50/50 demo tail mixing, HBM-resident MemoryEfficientReplayBuffer",
                   "batch_per_gpu": BATCH, "global_batch": gb, "utd": UTD,
                   "step_definition": f"1 step = 1 update() call = {UTD} critic grad-steps of {gb} rows + 1 "
                                      f"actor/temperature step",
                   "value_definition": "critic grad-steps/s x global_batch/256 (256-row grad-step equivalents)",
                   "parallelism": f"dp{world}"},
        "matmul_precision": precision,
        "model_tflops": round(value * FLOP_PER_GRAD_STEP / 1e12, 2),
        "highest_value": None if highest_value is None else round(highest_value, 3),
        "highest_note": "the same update() calls at precision 'highest' (exact fp32 MFMA), 3 timed steps in this run",
        "host_index_sampling_ms_per_update": round(sample_ms, 2),
        "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 2),
        "host_cpu_ms_per_step": round(host_cpu / args.steps * 1e3, 2),
        "host_note": "enqueue = wall time of the update() calls (includes waiting on a full hardware queue); cpu = "
                     "time.thread_time() of the calling thread over the same calls",
        "end_to_end_value": None if rehearsal else round(args.steps * UTD * world / e2e * (BATCH / 256.0), 3),
        "end_to_end_note": "same steps with buffer.sample (index draw + rejection + H2D of state/action leaves) inside "
                           "the loop; frames stay in HBM",
        "last_info": {k: round(float(v), 5) for k, v in info.items()},
    }
    if roofline is not None:
        out["roofline"] = roofline
    if rehearsal:
        out["rehearsal"] = True        # ranks share ONE GPU over gloo: a launch-path check, not a scaling measurement
        out["metric"] = "REHEARSAL (all ranks on one GPU, gloo) of: " + out["metric"]
    if world > 1:
        out["ranks_seen"] = world
        # time inside the gradient all-reduces of the timed region (stream events around each RCCL call; host wall time
        # under gloo), max over ranks: includes waiting for the slowest rank to arrive
        out["allreduce_ms_per_step"] = round(max_over_ranks([allreduce_s])[0] / args.steps * 1e3, 3)
    if rank == 0 and world == 1 and headline and not args.no_extra_configs:
        del agent, batches, it, buf
        torch.cuda.empty_cache()
        out.update(extra_configs(progress))
    if rank == 0 and world == 1 and not args.no_cpu_baseline and headline:
        progress("CPU baseline (oracle on the host cores, ~60 s) ...")
        out["cpu_baseline"] = cpu_baseline(progress=progress)
        out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        import torch.distributed as td
        td.destroy_process_group()


if __name__ == "__main__":
    main()
