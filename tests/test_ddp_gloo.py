"""GradSync (hyperpri_amd/ddp.py) over gloo, world_size 2, on CPU: averaged gradients must equal the mean
of the per-rank gradients, bucket by bucket, and be identical on both ranks."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hyperpri_amd.ddp import GradSync
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.ReLU(), torch.nn.Linear(33, 5), torch.nn.Linear(5, 1))
    sync = GradSync(net, bucket_mb=0.0001)   # ~100 B buckets -> several buckets
    assert len(sync.buckets) >= 3
    res = []
    for step in range(2):
        torch.manual_seed(100 + rank + 10 * step)
        x = torch.randn(4, 7)
        for p in net.parameters():
            p.grad = None
        net(x).sum().backward()
        sync.finish()
        res.append([p.grad.clone() for p in net.parameters()])
        # local (un-averaged) gradient for the check
        for p in net.parameters():
            p.grad = None
        sync2_local = torch.autograd.grad(net(x).sum(), list(net.parameters()))
        res.append([g.clone() for g in sync2_local])
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_world2():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    for step in range(2):
        avg0, loc0 = r0[2 * step], r0[2 * step + 1]
        avg1, loc1 = r1[2 * step], r1[2 * step + 1]
        for a0, a1, l0, l1 in zip(avg0, avg1, loc0, loc1):
            assert torch.equal(a0, a1)
            torch.testing.assert_close(a0, (l0 + l1) / 2, rtol=1e-6, atol=1e-7)


def test_gradsync_single_process_is_identity():
    from hyperpri_amd.ddp import GradSync
    net = torch.nn.Linear(3, 2)
    sync = GradSync(net)
    x = torch.randn(5, 3)
    net(x).sum().backward()
    g = [p.grad.clone() for p in net.parameters()]
    sync.finish()
    for a, b in zip(g, [p.grad for p in net.parameters()]):
        assert torch.equal(a, b)


def _worker_accum(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hyperpri_amd.ddp import GradSync
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.ReLU(), torch.nn.Linear(33, 1))
    sync = GradSync(net, bucket_mb=0.0001)
    xs = []
    for k in range(3):
        torch.manual_seed(200 + 10 * k + rank)
        xs.append(torch.randn(4, 7))
    local = [torch.zeros_like(p) for p in net.parameters()]
    for x in xs:
        for a, g in zip(local, torch.autograd.grad(net(x).sum(), list(net.parameters()))):
            a += g
    # two backward passes per step without no_sync(): must raise, not mix reduced and local gradients
    net(xs[0]).sum().backward()
    raised = False
    try:
        net(xs[1]).sum().backward()
    except RuntimeError as e:
        raised = "second backward" in str(e)
    sync.finish()
    # three micro-batches: two under no_sync, the last one communicates the sum
    for p in net.parameters():
        p.grad = None
    with sync.no_sync():
        for x in xs[:2]:
            net(x).sum().backward()
            sync.finish()
    net(xs[2]).sum().backward()
    sync.finish()
    out[rank] = (raised, [p.grad.clone() for p in net.parameters()], local)
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_world2_micro_batches_and_guard():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_accum, args=(world, _free_port(), out), nprocs=world, join=True)
    (r0, g0, l0), (r1, g1, l1) = out[0], out[1]
    assert r0 and r1
    for a0, a1, x0, x1 in zip(g0, g1, l0, l1):
        assert torch.equal(a0, a1)
        torch.testing.assert_close(a0, (x0 + x1) / 2, rtol=1e-5, atol=1e-6)


def test_bucket_plan_keeps_the_last_bucket_small():
    """The all-reduce of the last bucket cannot overlap with backward: the gradients that land last form a small tail."""
    from hyperpri_amd.ddp import GradSync
    net = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.Linear(64, 2048), torch.nn.Linear(2048, 1024), torch.nn.Linear(1024, 1024),
                              torch.nn.Linear(1024, 64), torch.nn.Linear(64, 1))
    sync = GradSync(net, bucket_mb=4.0, tail_mb=1.0)
    try:
        sizes = [b.flat.numel() * 4 for b in sync.buckets]
        assert len(sizes) >= 3 and sizes[-1] <= 1 << 20
        # ready order = reverse registration order, every parameter exactly once
        flat = [id(p) for b in sync.buckets for p in b.params]
        assert flat == [id(p) for p in reversed(list(net.parameters()))]
        # the tail holds the first layers (whose gradients land last)
        assert id(net[0].weight) in {id(p) for p in sync.buckets[-1].params}
    finally:
        sync.remove()


def _buf_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hyperpri_amd.ddp import GradSync
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    with torch.no_grad():
        net[1].running_mean.fill_(float(rank + 1))          # the ranks' statistics have drifted apart
    sync = GradSync(net, broadcast_buffers="train")         # training-mode forwards only (True = every forward, as torch DDP)
    net.eval()
    if rank == 0:
        net(torch.randn(3, 4))                              # validation on rank 0 only: an eval forward enters no collective
    assert torch.equal(net[1].running_mean, torch.full((4,), float(rank + 1)))
    net.train()
    with torch.no_grad():
        net[1].momentum = 0.0                               # keep the statistics as broadcast
        net(torch.randn(3, 4))                              # the forward pre-hook broadcasts rank 0's buffers first (ONE flat
    out[rank] = net[1].running_mean.clone()                 # tensor per dtype: float statistics, int64 counters)
    out[10 + rank] = int(net[1].num_batches_tracked)
    sync.remove()
    dist.barrier()
    dist.destroy_process_group()


def _buf_worker_eval(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hyperpri_amd.ddp import GradSync
    net = torch.nn.Sequential(torch.nn.Linear(4, 4), torch.nn.BatchNorm1d(4))
    with torch.no_grad():
        net[1].running_mean.fill_(float(rank + 1))
    sync = GradSync(net, broadcast_buffers=True)            # torch DDP's behaviour: EVERY forward, eval included
    net.eval()
    net(torch.randn(3, 4))                                  # validation on all ranks: rank 0's statistics are the ones in use
    out[rank] = net[1].running_mean.clone()
    sync.remove()
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_buffers_true_covers_eval_forwards():
    """with broadcast_buffers=True torch DDP broadcasts on every forward through the wrapper, whatever the mode."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_buf_worker_eval, args=(2, _free_port(), out), nprocs=2, join=True)
    assert torch.equal(out[0], torch.full((4,), 1.0)) and torch.equal(out[1], out[0])


def test_broadcast_buffers_like_torch_ddp():
    """GradSync(broadcast_buffers="train"): rank 0's BN buffers replace everyone's at the start of a training-mode forward, as one
    coalesced broadcast per dtype; eval-mode forwards do not communicate (rank-0-only validation)."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_buf_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert torch.equal(out[0], torch.full((4,), 1.0)) and torch.equal(out[1], out[0])
    assert out[10] == out[11] == 1
