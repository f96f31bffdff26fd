"""Host-side cost of update(): cProfile over a few calls (the GPU runs behind; we only look at Python time)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import bench
from helpers import pixel_spaces
from serl_amd.agents import DrQLearner
from serl_amd.data import MemoryEfficientReplayBuffer
obs_space, act_space = pixel_spaces(bench.KEYS, bench.H, bench.W, bench.STACK, bench.STATE_DIM, bench.ACT_DIM)
agent = DrQLearner.create(42, obs_space, act_space, pixel_keys=bench.KEYS, **bench.AGENT_KW)
buf = MemoryEfficientReplayBuffer(obs_space, act_space, 3000, pixel_keys=bench.KEYS, device="cuda")
bench.fill_buffer(buf, 2000, seed=0); buf.seed(0)
it = buf.get_iterator(queue_size=1, sample_args=dict(batch_size=bench.BATCH * bench.UTD, pack_obs_and_next_obs=True,
                                                     demo_batch_size=bench.BATCH * bench.UTD // 2, demo_size=500))
batches = [next(it) for _ in range(4)]
agent, _ = agent.update(batches[0], bench.UTD, pixel_keys=bench.KEYS)
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for b in batches[1:]:
    agent, _ = agent.update(b, bench.UTD, pixel_keys=bench.KEYS)
pr.disable()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue ms/update (with cProfile overhead)", (t1 - t0) * 1e3 / 3, " drain ms", (t2 - t1) * 1e3)
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
