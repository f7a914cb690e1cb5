"""``python bench.py --gpus N`` must start its own ranks (the driver's command shape has no launcher around it).
CPU rehearsal of that path: the parent spawns torch.distributed.run as a child, the ranks talk over gloo
(``--dry-launch`` replaces the GPU workload by a trivial all-reduce), rank 0's JSON line is relayed, and a failing
rank turns into a non-zero exit code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)


def test_bench_self_launches_two_ranks_and_relays_one_json_line():
    r = _run()
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["allreduce_sum"] == 3.0      # ranks contributed 1 + 2


def test_bench_self_launch_propagates_a_failing_rank():
    r = _run({"HPRI_DRY_LAUNCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
