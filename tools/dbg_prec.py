import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
import serl_amd
from helpers import *
from helpers import _leaves
from oracle import learner as ol
from serl_amd.agents import SACLearner
for prec in ("highest", "bf16_3x"):
    serl_amd.set_matmul_precision(prec)
    obs_space, act_space = state_spaces(17, 6)
    kw = dict(num_qs=2, critic_layer_norm=False, backup_entropy=True, init_temperature=1.0)
    agent = SACLearner.create(42, obs_space, act_space, **kw)
    ostate, ocfg = agent_to_oracle(agent)
    rng = np.random.default_rng(0)
    batch = make_state_batch(rng, 256, 17, 6)
    c64 = []
    ol.sac_update(ostate, ocfg, batch, 1, dtype=torch.float64, collect=c64)
    agent2, info = agent.update(batch, 1)
    ws = agent._ws(256)
    aux = c64[-2]
    print(prec, {k: (type(v).__name__, getattr(v, "shape", None)) for k, v in aux.items() if k != "grads"})
    tq = ws.tensor("target_q", (256,)).cpu().numpy()
    ref = np.asarray(aux["target_q"], dtype=np.float64).reshape(-1)
    print("  target_q abs err max %.2e  (scale %.2e)" % (np.abs(tq - ref).max(), np.abs(ref).max()))
    na = np.asarray(aux["next_actions"], dtype=np.float64)
    print("  next_actions scale", np.abs(na).max())
    for nm in ("q_C", "qs", "q"):
        try:
            t = ws.tensor(nm, (2, 256)).cpu().numpy(); print("  ws", nm, np.abs(t - np.asarray(aux["qs"], dtype=np.float64)).max())
        except Exception as e:
            pass
    print("  ws names:", [k for k in ws._bufs][:60] if hasattr(ws, "_bufs") else dir(ws))
