"""The second-stream policy of engine.py is decided at import time from the environment (no GPU needed): under a gradient
sink the second stream is used only when the HIP runtime will have 8 hardware queues."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = ("import os, sys; sys.path.insert(0, %r); from hyperpri_amd import engine; "
        "print(int(engine.SIDE_STREAM), int(engine.SIDE_STREAM_WITH_SINK), os.environ.get('GPU_MAX_HW_QUEUES'))" % ROOT)


def _run(env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "HPRI_SIDE_STREAM", "HPRI_SIDE_STREAM_SINK")}
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    return out.stdout.strip().split()


def test_second_stream_policy_from_environment():
    assert _run({}) == ["1", "1", "8"]                                  # nothing set: the package raises the queue count itself
    assert _run({"GPU_MAX_HW_QUEUES": "4"}) == ["1", "0", "4"]          # the caller pinned 4 queues: one stream under a sink
    assert _run({"GPU_MAX_HW_QUEUES": "16"}) == ["1", "1", "16"]
    assert _run({"HPRI_SIDE_STREAM_SINK": "0"})[:2] == ["1", "0"]       # explicit override
    assert _run({"HPRI_SIDE_STREAM": "0"})[0] == "0"
