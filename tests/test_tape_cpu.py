"""The reverse-mode tape's hand-over of parameter gradients to a gradient sink (engine.Tape.backward): a parameter that
several recorded ops contribute to (tied weights, a module called twice inside one fused tape) is handed over only after
the LAST of them has run -- a bucket's all-reduce must not start while a later node still accumulates into its slot.
Host logic only: fake nodes, a fake sink, CPU tensors."""
import torch

from hyperpri_amd import engine


class _Sink:
    def __init__(self, params):
        self.store = {id(p): torch.zeros_like(p) for p in params}
        self.order = []
        self.seen_at_ready = {}

    def slot(self, p):
        return self.store[id(p)], False

    def ready(self, p):
        self.order.append(id(p))
        self.seen_at_ready[id(p)] = self.store[id(p)].clone()


def test_shared_parameter_is_handed_over_after_its_last_node():
    w_shared = torch.nn.Parameter(torch.zeros(3))
    w_single = torch.nn.Parameter(torch.zeros(3))
    sink = _Sink([w_shared, w_single])
    engine.set_grad_sink(sink)
    try:
        tape = engine.Tape(True)

        def node(params, value):
            def bwd(tp):
                for p in params:
                    g, acc = tp.param_slot(p)
                    if acc:
                        g += value
                    else:
                        g.fill_(value)
            tape.note_params(*params)
            tape.nodes.append(bwd)
        node([w_shared], 1.0)               # forward order: shared, single, shared again
        node([w_single], 5.0)
        node([w_shared], 2.0)
        tape.backward()
    finally:
        engine.set_grad_sink(None)
    # backward order: second use of the shared weight (not final yet), the single-use weight, first use of the shared weight
    assert sink.order == [id(w_single), id(w_shared)]
    assert torch.equal(sink.seen_at_ready[id(w_shared)], torch.full((3,), 3.0))     # both contributions were in
    assert torch.equal(sink.seen_at_ready[id(w_single)], torch.full((3,), 5.0))


def test_unannounced_parameter_counts_as_single_use():
    w = torch.nn.Parameter(torch.zeros(2))
    sink = _Sink([w])
    engine.set_grad_sink(sink)
    try:
        tape = engine.Tape(True)
        tape.nodes.append(lambda tp: tp.param_slot(w)[0].fill_(1.0))
        tape.backward()
    finally:
        engine.set_grad_sink(None)
    assert sink.order == [id(w)]


def test_throttle_is_a_no_op_on_cpu_and_parses_its_bound():
    engine.throttle(torch.device("cpu"))
    assert engine.STEPS_IN_FLIGHT >= 0


def test_parameter_whose_last_announcing_node_returns_early_is_still_handed_over():
    """a module called twice in one tape with one result unused -- that node leaves before it asks for the slot.  The
    gradient the other call wrote into the bucket must reach ``ready()`` at the end of backward (GradSync.finish() zeroes slots
    that never landed)."""
    w = torch.nn.Parameter(torch.zeros(3))
    sink = _Sink([w])
    engine.set_grad_sink(sink)
    try:
        tape = engine.Tape(True)
        tape.note_params(w)
        tape.nodes.append(lambda tp: None)                               # first recorded use: its output got no gradient
        tape.note_params(w)
        tape.nodes.append(lambda tp: tp.param_slot(w)[0].fill_(4.0))     # second use (runs first in backward)
        tape.backward()
    finally:
        engine.set_grad_sink(None)
    assert sink.order == [id(w)]
    assert torch.equal(sink.seen_at_ready[id(w)], torch.full((3,), 4.0))


def test_bf16_gradient_storage_guards():
    """Round 4: activation gradients may be stored as bf16 rows (skips, single-reader tensors, the plane-GEMM path).  Host logic only: a
    channel slice of bf16 rows is bf16 rows; Tape.grad_slot hands an existing bf16 gradient only to a caller that says it can add
    into bf16 (the pooling backward) -- an fp32 writer meeting one is an internal error, not silent corruption; a bf16 view never
    meets an earlier gradient of the same tensor."""
    import pytest
    buf = torch.zeros(2 * 3 * 4 * 16, dtype=torch.bfloat16)
    g = engine.Act(buf, 2, 3, 4, 16, 16, 0)
    g.b16, g.f32_valid = True, False
    v = g.slice(8, 8)
    assert v.b16 and not v.f32_valid and v.coff == 8 and v.cs == 16 and v.buf is buf
    f = engine.Act(torch.zeros(2 * 3 * 4 * 16), 2, 3, 4, 16, 16, 0)
    assert not f.slice(4, 8).b16 and f.slice(4, 8).f32_valid
    x = engine.Act(torch.zeros(8), 2, 3, 4, 16, 16, 0)
    tape = engine.Tape(True)
    tape.grads[id(x)] = g
    got, acc = tape.grad_slot(x, b16_ok=True)
    assert got is g and acc
    with pytest.raises(RuntimeError, match="bf16 rows"):
        tape.grad_slot(x)
    with pytest.raises(RuntimeError, match="bf16 gradient view"):
        tape.set_grad_view(x, v)
    y = engine.Act(torch.zeros(8), 2, 3, 4, 8, 8, 0)
    tape.set_grad_view(y, v)                               # first gradient of y: the view is taken as it is
    assert tape.grads[id(y)] is v
