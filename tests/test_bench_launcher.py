"""``python bench.py --gpus N`` must start its own ranks (the driver's command shape has no launcher around it).
CPU rehearsal of that path: the parent spawns torch.distributed.run as a child, the ranks talk over gloo
(``--dry-launch`` replaces the GPU workload by a trivial all-reduce), rank 0's JSON line is relayed, and a failing
rank turns into a non-zero exit code."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env=None, gpus=2):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(extra_env or {})
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--dry-launch"],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)


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


def test_bench_self_launch_at_the_target_width_of_eight_ranks():
    """The driver's N = 8 command shape, rehearsed on CPU: eight gloo ranks, one JSON line, rank-dependent inputs (the sample
    seeds 1234 + rank * batch + i of the real workload are derived the same way), `rccl_ranks`-shaped bookkeeping."""
    r = _run(gpus=8)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["allreduce_sum"] == 36.0     # 1 + 2 + ... + 8
    assert out["ranks"] == 8 and out["global_batch"] == 16 and out["sample_seeds_last_rank"] == [1234 + 14, 1234 + 15]
