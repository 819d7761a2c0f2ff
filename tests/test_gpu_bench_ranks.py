"""`python bench.py --gpus 2` with no outer launcher starts two ranks itself (both on the one GPU of the test
box, gloo standing in for RCCL) and reports the size of the process group it really ran with."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg2g", "--envs", "64", "--steps", "2",
           "--warmup", "1", "--rollout-envs", "128", "--rollout-steps", "4"] + extra
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-4000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks():
    two = _run(["--gpus", "2"], {"FLM_DIST_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["dist_backend"] == "gloo"
    assert two["config"]["envs_per_gpu"] == 64 and two["scaling"] == "weak"
    steps = two["env_steps"]
    assert "error" not in steps and steps["config"]["n_gpus"] == 2
    assert steps["act_only"]["value"] > 0 and steps["act_and_learn"]["value"] > 0
    assert steps["act_and_learn"]["learn_steps_per_step"] > 0          # the gradient exchange really ran
    assert "cpu_baseline" not in two                                    # timed on rank 0 of a single-rank run only

    one = _run(["--gpus", "1"])
    assert one["n_gpus"] == 1 and one["dist_backend"] is None
    assert one["cpu_baseline"]["value"] > 0 and one["roofline"]["frac"] > 0
    # both ranks' environments are counted: the whole-job figure is per-rank work x ranks / slowest rank
    assert two["value"] > 0 and one["value"] > 0
