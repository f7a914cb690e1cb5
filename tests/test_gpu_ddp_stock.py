"""Stock torch DistributedDataParallel around the networks -- the reference's own N > 1 path (Lightning strategy="ddp",
PLTrainer.py:434-442) -- gets its buckets DURING backward: under a process group the network's tape runs as a chain of autograd
nodes (hyperpri_amd/autograd.py: run_staged), each handing its parameters' gradients to autograd when its slice of the backward
has been enqueued.  Child processes (tests/_ddp_stock_rank.py) wrap the network in DDP, register a communication hook that records
every bucket's position among the C-ABI launches of the backward and a HIP event, and store gradients for the comparison with
the one-node tape.  Needs a real MI355X: ``-m gpu``."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import record_margin

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _launch(tmp_path, world, backend, size, prec="fp32"):
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", GPU_MAX_HW_QUEUES="8")
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ddp_stock_rank.py"), str(r), str(world), port, outs[r],
                               backend, size, prec], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    return [np.load(o) for o in outs]


def _check_hand_over(z, tag):
    """All buckets but the last reach the communication hook before the first layer's weight gradient (the last weight-gradient
    launch of the backward) is enqueued; returns (fraction of gradient bytes handed over by then, per-bucket table)."""
    pos, nbytes, last = z["bucket_pos"], z["bucket_bytes"], z["bucket_is_last"]
    assert len(pos) >= 3 and int(last.sum()) == 1 and int(last[-1]) == 1, (pos, last)
    assert list(pos) == sorted(pos)
    assert len(z["plan"]) >= 4, z["plan"]                         # the tape really ran as a chain
    early = pos[:-1]
    assert int(early.max()) < int(z["last_wgrad_pos"]), (early, int(z["last_wgrad_pos"]))
    frac_bytes = float(nbytes[:-1].sum()) / float(nbytes.sum())
    record_margin(f"ddp_stock/{tag}/bytes_handed_over_before_first_conv_wgrad", 1.0 - frac_bytes, 0.2)
    return frac_bytes


def test_one_rank_rccl_full_size_buckets_leave_during_backward(tmp_path):
    """CubeNET-64 on two 608x968x238 cubes, DDP over RCCL (world 1), default 25 MiB buckets."""
    z = _launch(tmp_path, 1, "nccl", "full")[0]
    frac_bytes = _check_hand_over(z, "full_rccl")
    assert list(z["plan"]) == [3, 4, 1, 1, 1, 1, 1, 6]             # stages are single convolutions; cuts every >= 8 MB from the end
    # most of the gradient bytes are with the reducer while a good part of the backward is still to run.  Stock DDP
    # re-buckets in arrival order with its 25 MiB cap: what is left for the last bucket is whatever arrives after the last cap
    # was reached (down3's first convolution + down2 + down1 + stem, 9.4 MB); the HIP-event clock says how much of the
    # backward's GPU time was still ahead at each hand-over.
    ms, total = z["bucket_ms"], float(z["backward_ms"])
    cum = np.cumsum(z["bucket_bytes"]) / float(z["bucket_bytes"].sum())
    left = 1.0 - ms / total
    # measured (profiles/r05_ddp_stock_buckets.json): cumulative bytes 1.7 / 24 / 69 / 92 / 100 % with 52 / 38 / 35 / 29 / 1 % of the
    # backward's GPU time still ahead.  The gradient bytes sit in the deep layers, whose backward lies in the middle of the pass.
    k65, k85 = int(np.argmax(cum >= 0.65)), int(np.argmax(cum >= 0.85))
    assert frac_bytes >= 0.85, (frac_bytes, z["bucket_bytes"])
    assert left[k65] >= 0.30, (cum, left)                         # >= 65 % of the bytes handed over with >= 30 % of backward left
    assert left[k85] >= 0.22, (cum, left)                         # >= 85 % of the bytes with >= 22 % left
    assert z["launches_after_bucket"][-2] > 20                    # ... and the bucket before the last with real work behind it
    record_margin("ddp_stock/full_rccl/backward_left_at_65pct_bytes", 1.0 - float(left[k65]), 0.7)
    record_margin("ddp_stock/full_rccl/backward_left_at_85pct_bytes", 1.0 - float(left[k85]), 0.78)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    import json
    with open(os.path.join(out, "ddp_stock_buckets.json"), "w") as f:
        json.dump({"what": "stock DistributedDataParallel (RCCL, world 1, 25 MiB buckets) around CubeNET-64 on 2 x 238x608x968: per bucket, in "
                           "hand-over order", "plan_stages_per_node": [int(v) for v in z["plan"]],
                   "bucket_mb": [round(float(b) / 2 ** 20, 2) for b in z["bucket_bytes"]], "cumulative_byte_fraction": [round(float(c), 4) for c in cum],
                   "backward_gpu_time_left": [round(float(v), 4) for v in left], "backward_ms": round(total, 3),
                   "launches_of_backward_still_to_enqueue": [int(v) for v in z["launches_after_bucket"]], "launches_in_backward": int(z["n_launches"])}, f, indent=1)
    # one rank: the averaged gradients ARE the plain ones, and chain == one node bit for bit
    assert np.array_equal(z["plain_logits_head"], z["segmented_logits_head"])
    for k in z.files:
        if k.startswith("g/"):
            assert np.array_equal(z[k], z["plain/" + k[2:]]), k


@pytest.mark.parametrize("prec", ["fp32", "bf16", "f16"])
def test_two_ranks_gloo_stock_ddp_equals_mean_of_plain_gradients(tmp_path, prec):
    """(f16: every node of the chain takes the loss scale out of its own parameters' gradients before it hands them over.)"""
    z = _launch(tmp_path, 2, "gloo", "tiny", prec)
    for r in range(2):
        _check_hand_over(z[r], f"tiny_gloo_{prec}")
        assert np.array_equal(z[r]["plain_logits_head"], z[r]["segmented_logits_head"])
    for k in z[0].files:
        if not k.startswith("g/"):
            continue
        # allreduce_hook divides by the world size (exact for 2) and sums: (a + b) / 2 in fp32, on both ranks
        mean = (z[0]["plain/" + k[2:]] / np.float32(2) + z[1]["plain/" + k[2:]] / np.float32(2)).astype(np.float32)
        assert np.array_equal(z[0][k], z[1][k]), k
        assert np.array_equal(z[0][k], mean), k
