"""bench.py as the driver runs it, at toy sizes: the JSON contract, and --gpus N starting its own ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=400):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_on_one_gpu():
    d = _run(["--workload", "sparse3d:16", "--steps", "2", "--warmup", "1", "--extras", "none"])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "GFLOP/s" and d["dtype"] == "f64"
    assert d["verified"]["passed"] is True and d["verified"]["value"] <= 1e-12
    assert d["roofline"]["bound"] in ("hbm", "mfma") and 0 < d["roofline"]["frac"] < 1
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["value"] > 0
    assert d["value_first_factorization"] > 0 and d["cold_first_call_ms"] > 0


def test_bench_starts_its_own_ranks():
    """python bench.py --gpus 2 with no launcher around it: the two ranks are started by bench.py itself
    (here over gloo, both on the one GPU of the box) and rank 0's line says n_gpus = 2."""
    d = _run(["--gpus", "2", "--workload", "sparse3d:14", "--steps", "1", "--warmup", "1", "--extras", "none",
              "--no-cpu-baseline"], {"BENCH_BACKEND": "gloo", "BENCH_DEVICE_GIB": "8"})
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["multi"]["cut_blocks"] >= 1 and d["multi"]["levels"] >= 2
    assert 0 < d["top_flops_share"] < 1 and d["speedup_bound"] >= 1
