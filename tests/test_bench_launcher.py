"""`python bench.py --gpus N` must start its own ranks (BASELINE configs[4] is invoked that way).  CPU rehearsal:
the launcher path with two gloo ranks and no kernels -- exactly one JSON line on stdout, n_gpus == n_ranks == 2."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_bench_starts_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--launcher-selftest"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["n_ranks"] == 2 and rec["steps"] == 3 and rec["allreduce_ok"]


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=120)
    assert out.returncode != 0 and "does not match WORLD_SIZE" in out.stderr
