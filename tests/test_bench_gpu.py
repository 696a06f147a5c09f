"""bench.py as the driver runs it (one GPU): the JSON line with its roofline object.  The roofline leg wraps the engine's GEMM
and weight-gradient launchers with timers -- a launcher that grows an argument must not break the default bench run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_default_bench_line_with_roofline_leg():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu", "--no-others"],
                       env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["unit"] == "frames/s" and line["dtype"] == "bf16"
    roof = line["roofline"]
    assert roof["bound"] in ("mfma", "hbm") and roof["achieved"] > 0 and 0 < roof["frac"] < 1 and roof["peak"] > 0
    assert "wgrad_gemm_k" in roof["kernel"] or "gather_gemm_k" in roof["kernel"]
    assert line["config"]["graphs_captured"] >= 1 and line["config"]["dist_world_size"] == 1
