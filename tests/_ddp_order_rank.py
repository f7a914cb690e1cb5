"""Child process of tests/test_gpu_ddp_sink.py::test_bucket_hand_over_is_ordered_behind_both_compute_streams (not collected by
pytest).    python tests/_ddp_order_rank.py <port> [precision]

One rank, GradSync(force=True): the collective is replaced by ``flat.mul_(2)`` on a stream of its own that is ordered only behind
the stream current at the call (ProcessGroupNCCL's contract).  Gradients must be exactly twice the plain-loop gradients."""
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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1])
    dist.init_process_group("gloo", rank=0, world_size=1)
    import hyperpri_amd as H
    from hyperpri_amd import ddp, engine
    from oracle import hyperpri_oracle as O        # inputs only: the counter-based generator

    def u(seed, shape):
        return torch.from_numpy(O._u(seed, int(np.prod(shape))).reshape(shape).copy())
    assert engine.SIDE_STREAM and engine.SIDE_STREAM_WITH_SINK, "the weight-gradient stream must be live under the sink for this test"
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    net = H.CubeNET(6, 1, first_depth=64, bilinear=False)
    shapes = OrderedDict((k, tuple(v.shape)) for k, v in net.state_dict().items())
    net.load_state_dict(O.synth_state_dict(shapes))
    net = net.to(dev).train()
    if len(sys.argv) > 2 and sys.argv[2] != "fp32":
        H.set_precision(net, sys.argv[2])       # (f16: the loss scale leaves each gradient on the stream that hands its bucket over)
    x = u(1235, (2, 1, 6, 144, 200)).to(dev)
    m = (u(4321, (2, 1, 144, 200)) > 0.9).float().to(dev)
    torch.nn.BCEWithLogitsLoss()(net(x), m).backward()
    torch.cuda.synchronize()
    plain = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    for p in net.parameters():
        p.grad = None
    sync = ddp.GradSync(net, bucket_mb=4.0, tail_mb=0.25, force=True)
    comm = torch.cuda.Stream(device=dev)
    calls = []

    class _Work:
        def wait(self):
            torch.cuda.current_stream(dev).wait_stream(comm)
            return True

    def fake_all_reduce(flat, op=None, group=None, async_op=False):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))         # what ProcessGroupNCCL orders the collective behind
        comm.wait_event(ev)
        with torch.cuda.stream(comm):
            flat.mul_(2.0)
        calls.append(flat.numel())
        return _Work()
    ddp.dist.all_reduce = fake_all_reduce
    for _ in range(3):
        for p in net.parameters():
            p.grad = None
        torch.nn.BCEWithLogitsLoss()(net(x), m).backward()
        sync.finish()
        torch.cuda.synchronize()
        for k, p in net.named_parameters():
            assert torch.equal(p.grad, plain[k] * 2.0), k
    assert len(calls) == 3 * len(sync.buckets) and sync.issued["tape"] == len(calls), (len(calls), dict(sync.issued))
    sync.remove()
    dist.destroy_process_group()
    print("ORDER-OK buckets", len(calls) // 3)


if __name__ == "__main__":
    main()
