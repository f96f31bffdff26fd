"""State-based configs of BASELINE.json (C1: SAC 17-dim obs, 2 critics, UTD 1; C2: RLPD 10/2 critics + LayerNorm,
UTD 20): these are launch-latency bound (1.2 GFLOP and 26 MB per grad-step), so the figure of merit is us per
grad-step against the per-launch floor (tools/launch_floor.py), not a roofline fraction."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from helpers import make_state_batch, state_spaces
from serl_amd.agents import SACLearner
from serl_amd.data.replay_buffer import to_device

obs_space, act_space = state_spaces(17, 6)
for name, kw, B, U in (("C1 SAC", dict(num_qs=2), 256, 1),
                       ("C2 RLPD", dict(num_qs=10, num_min_qs=2, critic_layer_norm=True, backup_entropy=False,
                                        init_temperature=0.1), 256, 20)):
    agent = SACLearner.create(42, obs_space, act_space, **kw)
    rng = np.random.default_rng(0)
    batches = [to_device(make_state_batch(rng, B * U, 17, 6), torch.device("cuda")) for _ in range(8)]
    for b in batches[:3]:
        agent, info = agent.update(b, U)
    torch.cuda.synchronize()
    n = 40 if U == 1 else 10
    t0 = time.perf_counter()
    for i in range(n):
        agent, info = agent.update(batches[i % 8], U)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: batch {B}, UTD {U}: {n * U / dt:8.1f} grad-steps/s, {dt / (n * U) * 1e6:7.1f} us per grad-step "
          f"({dt / n * 1e3:.2f} ms per update call), last info {dict(info)}")
