"""bench.py as the driver runs it: one plain command per GPU count, no launcher."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_without_a_launcher_on_one_gpu():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the script starts both ranks itself (fresh child processes),
    they decode disjoint prompt shards and rank 0's line carries the job's totals.  Rehearsed on a one-GPU box with
    both ranks on device 0 and a gloo process group (KG_BENCH_DEVICE / KG_BENCH_BACKEND)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(KG_BENCH_BACKEND="gloo", KG_BENCH_DEVICE="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "small", "--steps", "2", "--warmup", "1",
           "--max-new-tokens", "48", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and j["value"] > 0
    assert j["config"]["parallelism"].startswith("prompt-sharded x2")
    assert j["lossless"].startswith("2/2")
    # one rank alone decodes its shard only: the two-rank job counted both shards
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--model", "small", "--steps", "2", "--warmup", "1",
                          "--max-new-tokens", "48", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert one.returncode == 0, one.stderr[-3000:]
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    assert j1["n_gpus"] == 1 and j1["roofline"]["frac"] > 0
