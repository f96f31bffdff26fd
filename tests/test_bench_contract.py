"""bench.py prints ONE JSON line with the fields the driver reads (contract in the task statement / DESIGN.md 4)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_contract(dev):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "grad-steps/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "bf16_3x" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 20 * 1000.0 / d["ms_per_step"]) <= 1e-2 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] >= 0.9 * r["algorithmic_bytes_per_launch"]
    # the other configurations the target is stated on (BASELINE configs 2, 4, 5), as extra keys
    assert d["c2_us_per_grad_step"] > 0 and d["c4_value"] > 0 and d["c5_1gpu_value"] > 0 and d["host_cpu_ms_per_step"] > 0
    assert 0 < d["c4_roofline"]["frac"] < 1


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal(dev):
    """`python bench.py --gpus 2` starts its own ranks (the command the driver runs for the scaling bench).  On a one-GPU
    box both ranks share cuda:0 over gloo (SERL_AMD_BENCH_REHEARSAL=1, tools/rehearse_dp.sh); a short workload.  Checks
    the launch path, the all-reduce hooks of the native driver and the aggregate value of the JSON line."""
    env = dict(os.environ, SERL_AMD_BENCH_REHEARSAL="1", REH_DUMP_AFTER="150", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--batch", "32", "--utd", "2"], capture_output=True, text=True, timeout=250, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["rehearsal"] is True and d["metric"].startswith("REHEARSAL") and d["ranks_seen"] == 2
    assert d["allreduce_ms_per_step"] > 0
    # 2 grad-steps per update on each of 2 ranks of 32 rows, normalised to 256-row batches
    assert abs(d["value"] - 2 * 2 * 1000.0 / d["ms_per_step"] * (32 / 256.0)) <= 1e-2 * d["value"]
