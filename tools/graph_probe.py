"""Timing experiment (tool, not product): does replaying one whole update() as a HIP graph beat enqueueing it?

The update is captured with torch.cuda.CUDAGraph around the unchanged `agent.update(...)` call (the driver's side
streams join the capture through their event waits), and the SAME update -- same agent, same batch, same keys: a
deterministic recomputation with identical results -- is replayed K times.  Compared with K eager calls of the same
update.  Prints ms per update for both, for the state config C2 and the pixel headline config C3.

  python tools/graph_probe.py [c2] [c3]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from helpers import make_state_batch, pixel_spaces, rollout_transitions, state_spaces
from serl_amd.data.replay_buffer import to_device


def run(name, agent, batch, upd, K):
    for _ in range(3):                      # warm-up: workspaces, lazily created events, kernel attributes
        a, info = upd(agent, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        a, info = upd(agent, batch)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / K
    ref = {k: float(v) for k, v in info.items()}
    ref_flat = a.critic.store.flat.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g.capture_begin(capture_error_mode="relaxed")
        ga, ginfo = upd(agent, batch)
        g.capture_end()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            g.replay()
        t_hostg = time.perf_counter() - t0
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / K
    same = torch.equal(ga.critic.store.flat, ref_flat)
    print(f"{name}: eager {eager * 1e3:8.3f} ms/update (host {t_host / K * 1e3:7.3f}),  graph replay {graph * 1e3:8.3f} ms/update "
          f"(host {t_hostg / K * 1e3:7.3f});  replayed result bit-identical to eager: {same}", flush=True)


which = sys.argv[1:] or ["c2", "c3"]
if "c2" in which:
    from serl_amd.agents import SACLearner
    obs_space, act_space = state_spaces(17, 6)
    agent = SACLearner.create(42, obs_space, act_space, num_qs=10, num_min_qs=2, critic_layer_norm=True,
                              backup_entropy=False, init_temperature=0.1)
    batch = to_device(make_state_batch(np.random.default_rng(0), 256 * 20, 17, 6), torch.device("cuda"))
    run("C2 state RLPD, batch 256, UTD 20", agent, batch, lambda a, b: a.update(b, 20), 20)
if "c3" in which:
    from serl_amd.agents import DrQLearner
    from serl_amd.data import MemoryEfficientReplayBuffer
    keys = ("wrist_1", "wrist_2")
    obs_space, act_space = pixel_spaces(keys, 128, 128, 3, 19, 7)
    agent = DrQLearner.create(42, obs_space, act_space, pixel_keys=keys, cnn_features=(32, 64, 128, 256),
                              cnn_filters=(3, 3, 3, 3), cnn_strides=(2, 2, 2, 2), latent_dim=64, num_qs=10, num_min_qs=2,
                              critic_layer_norm=True, backup_entropy=False, init_temperature=0.1)
    buf = MemoryEfficientReplayBuffer(obs_space, act_space, 3000, pixel_keys=keys, device="cuda")
    for tr in rollout_transitions(np.random.default_rng(0), 2000, keys, 128, 128, 3, 19, 7, episode_len=100):
        buf.insert(tr)
    buf.seed(0)
    batch = next(buf.get_iterator(queue_size=1, sample_args=dict(batch_size=5120, pack_obs_and_next_obs=True)))
    run("C3 pixel DrQ-RLPD, batch 256, UTD 20", agent, batch, lambda a, b: a.update(b, 20, pixel_keys=keys), 6)
