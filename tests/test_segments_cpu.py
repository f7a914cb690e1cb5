"""Segmented tapes (autograd.run_staged): a network's tape program cut into a chain of autograd nodes that share ONE tape, so
that stock DistributedDataParallel (Lightning strategy="ddp", PLTrainer.py:434-442) receives the parameter gradients of the
last stages while the backward of the first stages still runs.  Host logic only: a fake staged program of CPU tensor ops (the
autograd bridge is taken through its ``_forward`` / ``_backward`` without the device guard, the CUDA check is patched out)."""
import pytest
import torch

from hyperpri_amd import autograd as A
from hyperpri_amd import engine as E


class _CpuFn(A._HipFn):
    @staticmethod
    def forward(ctx, program, n_in, params, grad_on, *tensors):
        return A._HipFn._forward(ctx, program, n_in, params, grad_on, *tensors)

    @staticmethod
    def backward(ctx, gout):
        return A._HipFn._backward(ctx, gout)


@pytest.fixture
def cpu_bridge(monkeypatch):
    monkeypatch.setattr(A, "_HipFn", _CpuFn)
    monkeypatch.setattr(A, "USE_DISPATCHER", False)
    monkeypatch.setattr(A, "_require_cuda", lambda t, what: None)
    monkeypatch.setattr(E, "_require_cuda", lambda t, what: None)
    monkeypatch.setattr(A, "join_side", lambda dev: None)


def _cl(n, c, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, h, w, c, generator=g).permute(0, 3, 1, 2)       # channels-last, C % 8 == 0: Act.from_tensor is zero-copy


def _staged_net(ws, log):
    """y0 = x*w0 ; y1 = y0*w1 ; y2 = y1*w2 ; out = (y2 + y0)*w3   (y0 is a skip: two consumers, its gradient is summed inside
    the tape's gradient table).  One stage per weight; every backward node logs its name."""
    def flat(a):
        return a.buf.permute(0, 2, 3, 1).reshape(-1) if a.buf.dim() == 4 else a.buf

    def scale(tape, x, w, name, skip=None):
        y = E.Act.new(x.N, x.H, x.W, x.C, x.buf.device)
        src = flat(x) if skip is None else flat(x) + flat(skip)
        y.buf.copy_(src * w.detach())
        if tape.record:
            def bwd(tp):
                log.append(name)
                g = tp.grads[id(y)]
                dw, acc = tp.param_slot(w)
                val = (flat(g) * src).sum().reshape(w.shape)
                dw.copy_(dw + val if acc else val)
                for t in (x, skip):
                    if t is None:
                        continue
                    gx, acc = tp.grad_slot(t)
                    gx.buf.copy_(gx.buf + flat(g) * w.detach() if acc else flat(g) * w.detach())
            tape.nodes.append(bwd)
        return y

    def prog(tape, a, need):
        y0 = scale(tape, a[0], ws[0], "s0")
        yield
        y1 = scale(tape, y0, ws[1], "s1")
        yield
        y2 = scale(tape, y1, ws[2], "s2")
        yield
        return scale(tape, y2, ws[3], "s3", skip=y0)
    return prog


def _run(mode, min_mb, monkeypatch, x):
    monkeypatch.setattr(A, "SEGMENT_MB", min_mb)
    torch.manual_seed(0)
    ws = [torch.nn.Parameter(torch.rand(1) + 0.5) for _ in range(4)]
    log = []
    for i, w in enumerate(ws):
        w.register_post_accumulate_grad_hook(lambda p, i=i: log.append(f"grad w{i}"))
    xin = x.clone().requires_grad_(True)
    out = A.run_staged(_staged_net(ws, log), [xin], [[w] for w in ws], mode)
    out.backward(_cl(*[out.shape[i] for i in (0, 1, 2, 3)], 9))      # (a channels-last gradient: Act.from_tensor takes it as it is)
    return out.detach().clone(), [w.grad.clone() for w in ws], xin.grad.clone(), log


def test_plan_counts_from_the_end_of_the_network():
    mk = lambda mb: [torch.empty(int(mb * (1 << 18)))]          # fp32: 2^18 elements per MB
    stages = [mk(0.7), mk(0.8), mk(3.4), mk(13.5), mk(54), mk(35), mk(8.8), mk(2.2), mk(0.5)]      # CubeNET-64's stages
    assert A.plan_segments(stages, 8 << 20) == [3, 1, 1, 1, 3]
    assert A.plan_segments(stages, 1 << 40) == [9]
    assert A.plan_segments(stages, 0) == [1] * 9
    assert A.plan_segments([], 8 << 20) == []


def test_chain_equals_one_node_and_hands_gradients_over_early(cpu_bridge, monkeypatch):
    x = _cl(2, 8, 3, 5, 1)
    out1, g1, gx1, log1 = _run(True, 0.0, monkeypatch, x)           # no process group: one node
    assert A.LAST_PLAN == [] or log1.index("grad w3") > log1.index("s0")
    out2, g2, gx2, log2 = _run("segmented", 0.0, monkeypatch, x)    # four nodes
    assert A.LAST_PLAN == [1, 1, 1, 1]
    assert torch.equal(out1, out2) and torch.equal(gx1, gx2)
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))
    # one node: every parameter gradient arrives after the whole tape has run; chain: stage k's gradient arrives before stage k-1 runs
    assert [e for e in log1 if e.startswith("s")] == ["s3", "s2", "s1", "s0"]
    assert max(log1.index(f"s{i}") for i in range(4)) < min(log1.index(f"grad w{i}") for i in range(4))
    for k in (3, 2, 1):
        assert log2.index(f"grad w{k}") < log2.index(f"s{k - 1}")
    assert log2[-1] == "grad w0"


def test_a_node_runs_one_stage_ahead_before_it_waits_for_the_weight_gradient_stream(cpu_bridge, monkeypatch):
    """With weight gradients on a second stream a node must make the main stream wait for that stream before its gradients
    leave; it first enqueues the stage in front of its slice, so the wait is covered (autograd._HipFn._backward)."""
    x = _cl(2, 8, 3, 5, 1)
    out1, g1, gx1, _ = _run(True, 0.0, monkeypatch, x)
    events = []
    monkeypatch.setattr(A, "_side_mark", lambda dev: events.append("mark") or len(events))
    monkeypatch.setattr(A, "_wait_mark", lambda dev, mark: events.append("wait"))
    monkeypatch.setattr(A, "SEGMENT_MB", 0.0)
    torch.manual_seed(0)
    ws = [torch.nn.Parameter(torch.rand(1) + 0.5) for _ in range(4)]
    for i, w in enumerate(ws):
        w.register_post_accumulate_grad_hook(lambda p, i=i: events.append(f"grad w{i}"))
    inner = _staged_net(ws, events)

    def prog(tape, a, need):
        tape.used_side = True               # (as if the stages had issued weight gradients on the second stream)
        return (yield from inner(tape, a, need))
    xin = x.clone().requires_grad_(True)
    out = A.run_staged(prog, [xin], [[w] for w in ws], "segmented")
    out.backward(_cl(*out.shape, 9))
    assert events == ["s3", "mark", "s2", "wait", "grad w3", "mark", "s1", "wait", "grad w2", "mark", "s0", "wait", "grad w1", "grad w0"]
    assert torch.equal(out.detach(), out1) and torch.equal(xin.grad, gx1)
    assert all(torch.equal(w.grad, g) for w, g in zip(ws, g1))


def test_a_gradient_added_after_its_segment_left_is_an_internal_error(cpu_bridge, monkeypatch):
    """A parameter listed under a stage behind its last tape node would leave with unfinished contents: refused."""
    monkeypatch.setattr(A, "SEGMENT_MB", 0.0)
    w0, w1 = torch.nn.Parameter(torch.ones(1)), torch.nn.Parameter(torch.ones(1))

    def prog(tape, a, need):
        def node(p):
            def bwd(tp):
                tp.param_slot(p)[0].fill_(1.0)
            tape.nodes.append(bwd)
        y = E.Act.new(a[0].N, a[0].H, a[0].W, a[0].C, a[0].buf.device)
        y.buf.zero_()
        node(w1)                     # stage 0 touches w1 ...
        yield
        node(w1)
        node(w0)
        return y
    out = A.run_staged(prog, [_cl(1, 8, 2, 2, 3)], [[w0], [w1]], "segmented")     # ... which is listed under stage 1
    with pytest.raises(RuntimeError, match="already handed to autograd"):
        out.backward(_cl(1, 8, 2, 2, 4))


def test_stage_count_mismatch_is_reported(cpu_bridge, monkeypatch):
    monkeypatch.setattr(A, "SEGMENT_MB", 0.0)
    w = [torch.nn.Parameter(torch.ones(1)) for _ in range(3)]

    def prog(tape, a, need):
        yield
        return a[0]
    with pytest.raises(RuntimeError, match="fewer stages"):
        A.run_staged(prog, [_cl(1, 8, 2, 2, 3)], [[p] for p in w], "segmented")


def test_frozen_parameters_or_no_grad_fall_back_to_one_node(cpu_bridge, monkeypatch):
    x = _cl(1, 8, 2, 2, 5)
    monkeypatch.setattr(A, "SEGMENT_MB", 0.0)
    ws = [torch.nn.Parameter(torch.ones(1)) for _ in range(4)]
    ws[1].requires_grad_(False)
    A.LAST_PLAN[:] = []
    out = A.run_staged(_staged_net(ws, []), [x], [[w] for w in ws], "segmented")
    assert A.LAST_PLAN == [] and out.requires_grad
    for w in ws:
        w.requires_grad_(True)
    with torch.no_grad():
        out = A.run_staged(_staged_net(ws, []), [x], [[w] for w in ws], "segmented")
    assert A.LAST_PLAN == [] and not out.requires_grad


def test_networks_list_every_parameter_under_one_stage():
    import hyperpri_amd as H
    from hyperpri_amd.models import _stage_params
    for net in (H.UNet(3, 1, bilinear=False), H.UNet(3, 1, bilinear=True), H.CubeNET(6, 1, 64, bilinear=False),
                H.CubeNET(6, 1, 128, bilinear=False), H.CubeNET(6, 1, 128, bilinear=True), H.SpectralUNET(10, 1, 4),
                H.SpectralUNET(10, 1, 4, bnorm=False)):
        st = _stage_params(net)
        ids = [id(p) for ps in st for p in ps]
        assert len(ids) == len(set(ids)) == len(list(net.parameters())) and len(st) in (9, 18)
