"""CPU: bench.py's own argument path.  `python bench.py --gpus 2` with no launcher in the environment must start two ranks
itself (before any GPU call), run the index broadcast between them and print ONE JSON line with n_gpus == 2 -- rehearsed here
with gloo and the test double tests/bench_double.py in place of the HIP library (LNR_BENCH_DOUBLE=1)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv):
    env = dict(os.environ, LNR_BENCH_DOUBLE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2])
def test_bench_gpus_flag_starts_that_many_ranks(n):
    out = run_bench("--gpus", str(n), "--steps", "3", "--warmup", "1", "--workload", "small", "--reads", "64")
    assert out["n_gpus"] == n and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["ranks_reported_by_backend"] == n
    assert out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["distinct_batches"] == 4
    if n > 1:
        assert out["config"]["index_broadcast_s"] is not None and out["config"]["index_bytes"] > 8000
    for k in ("metric", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline"):
        assert k in out
