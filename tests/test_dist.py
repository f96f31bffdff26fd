"""Multi-rank path.  CPU (gloo, world_size 2): the data-parallel construction -- contiguous UTD slices sharded
by rank, noise keyed by the global row index, per-shard loss scaled by the global count, gradient sum by
all-reduce -- reproduces the single-process gradient (oracle as the compute).  GPU: the product learner
with 2 ranks on the one visible GPU (gloo moves CUDA tensors) equals the single-rank run on the global batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _init(rank, world, port):
    import torch.distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    td.init_process_group("gloo", rank=rank, world_size=world)


def _oracle_state():
    from oracle import learner as ol, prng
    from serl_amd.networks.spec import NetConfig, actor_layout, critic_layout, init_tree
    cfg = NetConfig(action_dim=2, num_qs=4, obs_dim=5, hidden_dims=(16, 16), critic_layer_norm=True)
    cp, ap = init_tree(critic_layout(cfg), [0]), init_tree(actor_layout(cfg), [1])
    z = lambda t: ol.tree_map(np.zeros_like, t)
    state = dict(rng=prng.prng_key(3), actor=dict(params=ap, mu=z(ap), nu=z(ap), count=0),
                 critic=dict(params=cp, mu=z(cp), nu=z(cp), count=0), target_critic=dict(params=cp),
                 temp=dict(params=dict(log_temp=np.float32(-1)), mu=dict(log_temp=np.float32(0)),
                           nu=dict(log_temp=np.float32(0)), count=0))
    ocfg = dict(discount=0.99, tau=0.005, target_entropy=-1.0, num_qs=4, num_min_qs=2, backup_entropy=True,
                actor_lr=3e-4, critic_lr=3e-4, temp_lr=3e-4, critic_layer_norm=True, pixel_keys=(), cnn_strides=())
    return state, ocfg


def _cpu_worker(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_state_batch
    from oracle import learner as ol
    from serl_amd.dist import DataParallel, shard_rows
    _init(rank, world, port)
    dp = DataParallel()
    state, ocfg = _oracle_state()
    bg, utd = 8, 2
    batch = make_state_batch(np.random.default_rng(0), bg * utd, 5, 2)
    rows = shard_rows(bg * utd, utd, rank, world)
    b = bg // world
    local = ol.tree_map(lambda a: a[rows], batch)
    mini = ol.tree_map(lambda a: a[:b], local)                         # first UTD slice of this rank
    _, _, aux = ol.update_critic(state, ocfg, mini, torch.float64, shard=(rank * b, bg))
    flat = torch.cat([torch.tensor(g).reshape(-1) for _, g in ol.tree_leaves(aux["grads"])])
    dp.all_reduce_sum(flat)
    if rank == 0:
        out.put(flat.numpy())
    import torch.distributed as td
    td.barrier()
    td.destroy_process_group()


def test_dp_gradient_equals_global_gloo():
    from helpers import make_state_batch
    from oracle import learner as ol
    from serl_amd.dist import shard_rows
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_cpu_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    state, ocfg = _oracle_state()
    batch = make_state_batch(np.random.default_rng(0), 16, 5, 2)
    mini = ol.tree_map(lambda a: a[:8], batch)
    _, _, aux = ol.update_critic(state, ocfg, mini, torch.float64)
    ref = np.concatenate([np.asarray(g).reshape(-1) for _, g in ol.tree_leaves(aux["grads"])])
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-8)
    # shard_rows partitions the global batch, UTD-slice-major
    r0, r1 = shard_rows(16, 2, 0, 2), shard_rows(16, 2, 1, 2)
    assert r0.tolist() == [0, 1, 2, 3, 8, 9, 10, 11] and sorted(np.concatenate([r0, r1]).tolist()) == list(range(16))


def _gpu_case(large):
    """small: the round-1 case; large: a global batch of 512 rows (2 x 256 per rank) through the E1 conv stack."""
    keys = ("wrist_1", "wrist_2")
    if large:
        return dict(keys=keys, H=64, W=64, S=3, bg=512, utd=2, kw=dict(
            cnn_features=(32, 64, 128, 256), cnn_filters=(3, 3, 3, 3), cnn_strides=(2, 2, 2, 2), latent_dim=64,
            num_qs=10, num_min_qs=2, critic_layer_norm=True, backup_entropy=False, init_temperature=0.1))
    return dict(keys=keys, H=32, W=32, S=2, bg=8, utd=2,
                kw=dict(num_qs=4, num_min_qs=2, critic_layer_norm=True, backup_entropy=True))


def _gpu_worker(rank, world, port, out, large=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import make_pixel_batch, pixel_spaces
    from oracle import learner as ol
    from serl_amd.agents import DrQLearner
    from serl_amd.dist import DataParallel, shard_rows
    _init(rank, world, port)
    torch.cuda.set_device(0)
    dp = DataParallel()
    c = _gpu_case(large)
    keys, H, W, S, bg, utd = c["keys"], c["H"], c["W"], c["S"], c["bg"], c["utd"]
    obs_space, act_space = pixel_spaces(keys, H, W, S, 19, 7)
    agent = DrQLearner.create(5, obs_space, act_space, pixel_keys=keys, dist=dp, **c["kw"])
    batch = make_pixel_batch(np.random.default_rng(2), bg * utd, keys, H, W, S, 19, 7, packed=True)
    rows = shard_rows(bg * utd, utd, rank, world)
    local = ol.tree_map(lambda a: a[rows], batch)
    new, info = agent.update(local, utd, pixel_keys=keys)
    if rank == 0:       # host copies only: a CUDA tensor on the queue would need the producer alive at get() time (IPC)
        out.put((dict(info), new.critic.store.flat.cpu().numpy(), new.actor.store.flat.cpu().numpy(),
                 agent._shared["critic_grads"].flat.cpu().numpy()))
    import torch.distributed as td
    td.barrier()
    td.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("large", [False, True])
def test_dp_learner_two_ranks_equals_one(dev, large):
    from helpers import make_pixel_batch, pixel_spaces
    from serl_amd.agents import DrQLearner
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, out, large)) for r in range(2)]
    for p in procs:
        p.start()
    info2, critic2, actor2, grads2 = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    c = _gpu_case(large)
    keys, H, W, S, bg, utd = c["keys"], c["H"], c["W"], c["S"], c["bg"], c["utd"]
    obs_space, act_space = pixel_spaces(keys, H, W, S, 19, 7)
    agent = DrQLearner.create(5, obs_space, act_space, pixel_keys=keys, **c["kw"])
    batch = make_pixel_batch(np.random.default_rng(2), bg * utd, keys, H, W, S, 19, 7, packed=True)
    new, info1 = agent.update(batch, utd, pixel_keys=keys)
    for k in info1:
        assert abs(info1[k] - info2[k]) <= 1e-3 * max(abs(info1[k]), 1e-2), (k, dict(info1), info2)
    g1 = agent._shared["critic_grads"].flat.cpu().numpy()
    assert np.abs(g1 - grads2).max() <= 2e-2 * np.abs(g1).max()       # last UTD step, after one Adam step of drift
    assert np.abs(new.critic.store.flat.cpu().numpy() - critic2).max() <= 2.5 * 3e-4 * 2
    assert np.abs(new.actor.store.flat.cpu().numpy() - actor2).max() <= 2.5 * 3e-4


class _PairDP:
    """Two emulated ranks in ONE process (two threads, two streams, one GPU) with the ordering contract of an RCCL
    all-reduce: the sum is ENQUEUED on the caller's current stream, ordered against the peer's producer / consumer
    kernels by events only -- no host-side wait for the GPU anywhere (the host barrier below only makes sure the peer
    has enqueued its work, as a collective's rendezvous does).  The gloo path of serl_amd.dist blocks on a host copy and
    therefore cannot show a missing stream dependency between all-reduce, gradient side streams and the optimizer step;
    this one does.  Sum order fixed (rank 0 + rank 1): both ranks hold the same bits."""

    class Shared:
        def __init__(self):
            import threading
            self.barrier = threading.Barrier(2, timeout=120)
            self.slot = [None, None]
            self.read = [None, None]
            self.calls = [0, 0]

    def __init__(self, rank, shared):
        self.rank, self.world_size, self.sh = rank, 2, shared

    def all_reduce_sum(self, t):
        sh, r, o = self.sh, self.rank, 1 - self.rank
        st = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(st)
        sh.slot[r] = (t, ready)
        sh.calls[r] += 1
        sh.barrier.wait()
        ot, oready = sh.slot[o]
        assert ot.numel() == t.numel(), "the ranks disagree on the bucket"
        st.wait_event(oready)                               # the peer's gradient kernels
        tmp = (t + ot) if r == 0 else (ot + t)
        done = torch.cuda.Event()
        done.record(st)
        sh.read[r] = done
        sh.barrier.wait()
        st.wait_event(sh.read[o])                           # the peer has read my buffer before I overwrite it
        t.copy_(tmp)
        sh.barrier.wait()
        return t


def _pair_case(large):
    c = _gpu_case(large)
    if large:
        c = dict(c, bg=128)
    return c


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", ["1", "0"])
@pytest.mark.parametrize("large", [False, True])
def test_dp_stream_ordered_pair_equals_one_rank(dev, large, overlap, monkeypatch):
    """world = 2 on one GPU with a stream-ordered, host-sync-free all-reduce (_PairDP): the two ranks' results equal the
    single-rank run on the global batch -- with the gradient in two buckets, the trunk + ensemble bucket all-reduced on the
    driver's comm stream next to the encoder backward (SERL_AMD_DP_OVERLAP=1, default), and with one all-reduce per step."""
    import threading
    from helpers import make_pixel_batch, pixel_spaces
    from oracle import learner as ol
    from serl_amd.agents import DrQLearner
    from serl_amd.dist import shard_rows
    monkeypatch.setenv("SERL_AMD_DP_OVERLAP", overlap)          # read when the native handle is created
    c = _pair_case(large)
    keys, H, W, S, bg, utd = c["keys"], c["H"], c["W"], c["S"], c["bg"], c["utd"]
    obs_space, act_space = pixel_spaces(keys, H, W, S, 19, 7)
    batch = make_pixel_batch(np.random.default_rng(2), bg * utd, keys, H, W, S, 19, 7, packed=True)
    shared = _PairDP.Shared()
    res, errs = [None, None], []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                agent = DrQLearner.create(5, obs_space, act_space, pixel_keys=keys, dist=_PairDP(rank, shared), **c["kw"])
                local = ol.tree_map(lambda a: a[shard_rows(bg * utd, utd, rank, 2)], batch)
                new, info = agent.update(local, utd, pixel_keys=keys)
                info = dict(info)
                torch.cuda.current_stream().synchronize()
                res[rank] = (info, new.critic.store.flat.clone(), new.actor.store.flat.clone(),
                             agent._shared["critic_grads"].flat.clone())
        except Exception as e:      # noqa: BLE001
            errs.append((rank, repr(e)))
            shared.barrier.abort()
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=200)
    assert not errs, errs
    torch.cuda.synchronize()
    # the ranks agree bit for bit (same summed gradients, same optimizer arithmetic)
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2]) and res[0][0] == res[1][0]
    # per critic step: 2 buckets (overlap) or 1 all-reduce, + actor gradient + info at the end
    assert shared.calls[0] == shared.calls[1] == (2 * utd if overlap == "1" else utd) + 2
    agent = DrQLearner.create(5, obs_space, act_space, pixel_keys=keys, **c["kw"])
    new, info1 = agent.update(batch, utd, pixel_keys=keys)
    info2 = res[0][0]
    for k in info1:
        assert abs(info1[k] - info2[k]) <= 2e-4 * max(abs(info1[k]), 1e-2), (k, dict(info1), info2)
    g1 = agent._shared["critic_grads"].flat
    assert float((g1 - res[0][3]).abs().max()) <= 2e-2 * float(g1.abs().max())      # last UTD step, after one Adam step of drift
    assert float((new.critic.store.flat - res[0][1]).abs().max()) <= 2.5 * 3e-4 * 2
    assert float((new.actor.store.flat - res[0][2]).abs().max()) <= 2.5 * 3e-4
