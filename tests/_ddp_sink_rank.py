"""One rank of tests/test_gpu_ddp_sink.py (started as a fresh child process; not collected by pytest).

    python tests/_ddp_sink_rank.py <rank> <world> <port> <out.npz> [precision]

Every rank computes on cuda:0 (the GPU box has one card); the collective runs over gloo, as RCCL refuses two ranks on one
device.  What is under test is the ENGINE-SINK path of GradSync: the whole network is one autograd node, its weight-gradient
kernels write into the flat buckets, and the tape hands each bucket to the collective while backward is still running."""
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    prec = sys.argv[5] if len(sys.argv) > 5 else "fp32"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import hyperpri_amd as H
    from hyperpri_amd import engine
    from hyperpri_amd.ddp import GradSync
    from oracle import hyperpri_oracle as O        # inputs only: the counter-based generator (the checker runs in the parent)

    def u(seed, shape):
        return torch.from_numpy(O._u(seed, int(np.prod(shape))).reshape(shape).copy())

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    net = H.CubeNET(6, 1, first_depth=64, bilinear=False)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    net.load_state_dict(O.synth_state_dict(shapes))
    net = net.to(dev).train()
    if prec != "fp32":
        H.set_precision(net, prec)
    assert net.fused_tape
    x = u(1235 + rank, (2, 1, 6, 36, 50)).to(dev)
    m = (u(4321 + rank, (2, 1, 36, 50)) > 0.9).float().to(dev)
    res = {}
    # this rank's gradients from the PLAIN loop (no sink, no collective), same HIP kernels: the parent holds the synchronised
    # gradients to the exact mean of the two ranks' plain gradients (the kernels are deterministic)
    torch.nn.BCEWithLogitsLoss()(net(x), m).backward()
    torch.cuda.synchronize()
    for k, p in net.named_parameters():
        res["plain/" + k] = p.grad.detach().cpu().numpy()
        p.grad = None
    sync = GradSync(net, bucket_mb=4.0, tail_mb=0.25)        # 31 M parameters -> a handful of buckets
    for step in range(2):                                    # the second step reuses the buckets the first one installed as .grad
        for p in net.parameters():
            p.grad = None
        loss = torch.nn.BCEWithLogitsLoss()(net(x), m)
        loss.backward()
        pending_before_finish = sum(1 for b in sync.buckets if b.work is None)
        sync.finish()
        torch.cuda.synchronize()
        res[f"loss{step}"] = float(loss)
        res[f"left_for_finish{step}"] = pending_before_finish
    for k, p in net.named_parameters():
        res["g/" + k] = p.grad.detach().cpu().numpy()
    ov = sync.overlap_ms() or {}
    res["buckets"] = len(sync.buckets)
    res["issued_tape"], res["issued_hook"], res["issued_finish"] = (sync.issued[k] for k in ("tape", "hook", "finish"))
    res["issue_to_finish_ms"] = np.asarray(ov.get("issue_to_finish_ms", []), dtype=np.float64)
    res["side_stream_with_sink"] = int(engine.SIDE_STREAM and engine.SIDE_STREAM_WITH_SINK)
    np.savez(out, **res)
    dist.barrier()
    sync.remove()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
