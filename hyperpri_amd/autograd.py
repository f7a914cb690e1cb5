"""torch.autograd bridge: one Function node per reference module call.

``run(program, inputs, params)`` executes ``program(tape, acts, need)`` -- a sequence of HIP ops from
``engine`` -- and returns an ordinary autograd-tracked tensor, as the reference's callers expect
(PLTrainer.py:85-88: BCEWithLogitsLoss(pred, mask), torch.sigmoid(pred.detach()), pred.cpu()).
Feature maps are returned as channels-last views (logical NCHW shape) so module-to-module hand-offs
never copy; logits are plain contiguous (N, n_classes, H, W).
Backward runs the tape in reverse on the autograd worker thread with the thread's current stream.
"""
from __future__ import annotations

import itertools
import os
import threading
from typing import Callable, List, Optional, Sequence

import torch

from . import _lib
from . import engine as _E
from .engine import Act, Tape, _require_cuda, join_side, scale_tensors_


def _as4d(t: torch.Tensor) -> torch.Tensor:
    if t.dim() == 5:           # CubeNET cube (N,1,D,H,W) -> (N,D,H,W); dataset.py:269-271
        if t.shape[1] != 1:
            raise RuntimeError("hyperpri_amd: 5-D input must be (N,1,D,H,W)")
        r = t.reshape(t.shape[0], t.shape[2], t.shape[3], t.shape[4])
        if getattr(t, "_hpri_zero_padded", False):      # ingest.py cubes: keep the zero-copy channels-last marker
            r._hpri_zero_padded = True
        return r
    if t.dim() != 4:
        raise RuntimeError(f"hyperpri_amd: expected a 4-D or 5-D tensor, got shape {tuple(t.shape)}")
    return t


# ---- segmented tapes ---------------------------------------------------------------------------------------------------------------
# A whole network as ONE autograd node hands all its parameter gradients to autograd at the very end of backward: stock
# DistributedDataParallel (Lightning strategy="ddp", PLTrainer.py:434-442) -- whose reducer sends a bucket when the AccumulateGrad
# nodes of its parameters have run -- would then start its first all-reduce after the last kernel of backward.  Under a process group
# the networks therefore run as a CHAIN of a few autograd nodes that share one tape: the program is a generator that yields at its
# stage boundaries (module by module), consecutive stages are grouped into segments of at least SEGMENT_MB of parameters (counted from
# the end of the network: the order backward reaches them), node k's forward advances the generator through its stages, node k's
# backward runs its slice of the shared tape and returns the gradients of ITS parameters.  The kernels, their order and the buffers
# (skip gradients accumulate inside the tape's own gradient table as before -- autograd never adds two activation gradients) are
# those of the one-node tape, so logits and gradients are bit-identical; between the nodes autograd carries a one-element token.
SEGMENT_MB = 8.0
_TOKEN = object()
_token_grads: dict = {}


def _token_grad(device) -> torch.Tensor:
    g = _token_grads.get(device)
    if g is None:
        g = _token_grads[device] = torch.zeros(1, dtype=torch.float32, device=device)
    return g


class _SharedTape:
    __slots__ = ("tape", "gen", "gen_fn", "bounds")

    def __init__(self, gen_fn):
        self.tape, self.gen, self.gen_fn = Tape(True), None, gen_fn
        self.bounds: List[int] = []        # tape length after each stage, in stage order


class _Segment:
    """The tape program of one node of the chain: advances the shared generator by ``nstages`` stages (``s0``: index of its
    first stage in the network's stage list)."""
    __slots__ = ("shared", "nstages", "s0", "last")

    def __init__(self, shared: _SharedTape, nstages: int, s0: int, last: bool):
        self.shared, self.nstages, self.s0, self.last = shared, nstages, s0, last

    @property
    def first(self) -> bool:
        return self.s0 == 0

    def ahead(self) -> int:
        """Tape index at which the stage in FRONT of this node's first stage begins: the node's backward runs on into that stage
        before it hands its gradients over (see _HipFn._backward)."""
        b = self.shared.bounds
        return b[self.s0 - 2] if self.s0 >= 2 else 0

    def __call__(self, tape, acts, need):
        sh = self.shared
        if self.first:
            sh.gen = sh.gen_fn(tape, acts, need)
        try:
            for _ in range(self.nstages):
                next(sh.gen)
                sh.bounds.append(len(tape.nodes))
        except StopIteration as stop:
            if not self.last or len(sh.bounds) != self.s0 + self.nstages - 1:
                raise RuntimeError("hyperpri_amd: internal error: the tape program has fewer stages than its stage list")
            sh.gen = None
            sh.bounds.append(len(tape.nodes))
            return stop.value
        if self.last:
            raise RuntimeError("hyperpri_amd: internal error: the tape program has more stages than its stage list")
        return _TOKEN


def drain(gen_fn: Callable) -> Callable:
    """A staged (generator) tape program as a plain one: all stages in one go."""
    def program(tape, acts, need):
        g = gen_fn(tape, acts, need)
        try:
            while True:
                next(g)
        except StopIteration as stop:
            return stop.value
    return program


def plan_segments(stage_params: Sequence[Sequence[torch.Tensor]], min_bytes: int) -> List[int]:
    """Stage counts of the chain's nodes, in forward order.  Cuts are placed walking the stages from the END of the network (the
    order backward finishes them): a segment is closed once it holds ``min_bytes`` of parameters; what is left at the front
    forms the segment whose gradients leave last."""
    counts, n, size = [], 0, 0
    for params in reversed(list(stage_params)):
        n += 1
        size += sum(p.numel() * p.element_size() for p in params)
        if size >= min_bytes:
            counts.append(n)
            n, size = 0, 0
    if n:
        counts.append(n)
    return counts[::-1]


def segmentation_wanted(mode) -> bool:
    """``fused_tape`` of a network: "segmented" forces the chain, True picks it whenever a process group exists and no GradSync
    gradient sink is installed (the sink overlaps from inside one node and needs no chain)."""
    if mode == "segmented":
        return True
    if mode is not True or not SEGMENT_AUTO:
        return False
    from . import engine
    import torch.distributed as dist
    return engine._GRAD_SINK is None and dist.is_available() and dist.is_initialized()


SEGMENT_AUTO = True                # False: a process group alone does not select the chain (``fused_tape = "segmented"`` still does)
RUN_AHEAD = True                   # see _HipFn._backward (False: every node joins the weight-gradient stream at the end of its slice)


def _side_mark(device):
    from .engine import _side
    ev = torch.cuda.Event()
    ev.record(_side(device))
    return ev


def _wait_mark(device, mark) -> None:
    torch.cuda.current_stream(device).wait_event(mark)
LAST_PLAN: List[int] = []          # stage counts of the most recent segmented forward (tests / bench)


def run_staged(gen_fn: Callable, inputs: Sequence[torch.Tensor], stages: Sequence[Sequence[torch.Tensor]], mode=True,
               input_planes: int = 0, name: str = "run_program", lib_kind: Optional[str] = None) -> torch.Tensor:
    """Run a staged tape program (a generator yielding after each of its ``len(stages)`` stages but the last; ``stages[i]`` = the
    parameters stage i reads) -- as one autograd node, or, when ``segmentation_wanted(mode)``, as a chain of nodes."""
    params = [p for st in stages for p in st]
    if not (torch.is_grad_enabled() and segmentation_wanted(mode) and all(p.requires_grad for p in params)):
        return run(drain(gen_fn), inputs, params, input_planes, name, lib_kind)
    counts = plan_segments(stages, int(SEGMENT_MB * (1 << 20)))
    LAST_PLAN[:] = counts
    if len(counts) < 2:
        return run(drain(gen_fn), inputs, params, input_planes, name, lib_kind)
    shared = _SharedTape(gen_fn)
    out, s0 = None, 0
    for k, n in enumerate(counts):
        seg = _Segment(shared, n, s0, last=(k == len(counts) - 1))
        seg_params = [p for st in stages[s0:s0 + n] for p in st]
        out = run(seg, list(inputs) if k == 0 else [out], seg_params, input_planes if k == 0 else 0, "segment", lib_kind)
        s0 += n
    return out


class _HipFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, program: Callable, n_in: int, params: Sequence[torch.Tensor], grad_on, *tensors: torch.Tensor):
        with torch.cuda.device(tensors[0].device):      # launches and current_stream() follow the data's device
            return _HipFn._forward(ctx, program, n_in, params, grad_on, *tensors)

    @staticmethod
    def _forward(ctx, program: Callable, n_in: int, params: Sequence[torch.Tensor], grad_on, *tensors: torch.Tensor):
        ctx.kind = grad_on[2] if len(grad_on) > 2 else None       # which library the tape's launches go to (_lib.using)
        with _lib.using(ctx.kind):
            return _HipFn._forward_impl(ctx, program, n_in, params, grad_on, *tensors)

    @staticmethod
    def _forward_impl(ctx, program: Callable, n_in: int, params: Sequence[torch.Tensor], grad_on, *tensors: torch.Tensor):
        grad_on, input_planes = grad_on[0], grad_on[1]          # (caller's grad mode, bf16 planes wanted for the inputs)
        inputs = tensors[:n_in]
        need = list(ctx.needs_input_grad[4:])
        # needs_input_grad is True for trainable parameters even under torch.no_grad() / inference_mode(), and
        # is_grad_enabled() is always False inside Function.forward: the caller's grad mode is passed in by run()
        record = grad_on and any(need)
        seg = program if isinstance(program, _Segment) else None
        if seg is not None:
            # one slice of a segmented network (run_segmented): the tape is shared by the chain's nodes; only the first node
            # sees the network's inputs, the others a token that carries nothing but the graph edge
            if not record:
                raise RuntimeError("hyperpri_amd: internal error: a tape segment without a recorded tape")
            tape = seg.shared.tape
            inputs = inputs if seg.first else ()
        else:
            tape = Tape(record)
        acts: List[Act] = []
        raw_ok = input_planes == _E.INPUT_RAW_OK       # the program's first operation may read the caller's NCHW tensor itself
        if raw_ok:
            input_planes = -1
        for t in inputs:
            _require_cuda(t, "input tensor")
            lazy = Act.raw_nchw(_as4d(t), input_planes) if (raw_ok and not record and seg is None) else None
            acts.append(lazy if lazy is not None else Act.from_tensor(_as4d(t), input_planes))
        lo = len(tape.nodes)
        out = program(tape, acts, need[:n_in])
        ctx.slice = (lo, len(tape.nodes), seg.ahead()) if seg is not None else None
        if out is _TOKEN:
            res = torch.empty(1, dtype=torch.float32, device=tensors[0].device)
            ctx.out_act, ctx.holder = None, None
        elif isinstance(out, Act):
            res = out.to_tensor()
            ctx.out_act, ctx.holder = out, None
        else:
            res, holder = out
            ctx.out_act, ctx.holder = None, holder
        if record:
            ctx.tape, ctx.acts, ctx.params, ctx.n_in = tape, acts, list(params), n_in
            ctx.in_shapes = [tuple(t.shape) for t in inputs]
            ctx.in_cl = [a.buf is _as4d(t) for a, t in zip(acts, inputs)]
        return res

    @staticmethod
    def backward(ctx, gout: torch.Tensor):
        with torch.cuda.device(gout.device):
            return _HipFn._backward(ctx, gout)

    @staticmethod
    def _backward(ctx, gout: torch.Tensor):
        with _lib.using(getattr(ctx, "kind", None)):
            return _HipFn._backward_impl(ctx, gout)

    @staticmethod
    def _backward_impl(ctx, gout: torch.Tensor):
        tape: Tape = ctx.tape
        if tape is None:
            raise RuntimeError("hyperpri_amd: backward called twice (retain_graph is not supported)")
        ctx.tape = None
        if ctx.out_act is not None:
            tape.grads[id(ctx.out_act)] = Act.from_tensor(gout)
        elif ctx.holder is not None:
            ctx.holder["g"] = gout
        if ctx.slice is not None:
            # (an inner segment's incoming gradient is its token's: nothing to read)
            lo, hi, ahead = ctx.slice
            tape.backward(lo, hi if tape.done_to is None else min(hi, tape.done_to))
            if lo > 0 and tape.used_side and RUN_AHEAD:
                # The gradients that leave with this node were written on the weight-gradient stream; whoever receives them (the
                # reducer's copy into its bucket) runs on the main stream, which must wait for that stream.  Waiting right here
                # would idle the main stream until the last weight gradient of the slice has finished (0.8 % of an fp32 step and
                # 1.4 % of a bf16 step over four cuts: profiles/r05_segment_ab.json) -- so the node first enqueues the stage in
                # FRONT of its slice (the next node's last stage) and only then makes the main stream wait for the point the
                # weight-gradient stream had reached when the slice ended: that stage's kernels cover the wait.
                dev = gout.device
                mark, nkeep = _side_mark(dev), len(tape.side_keep)
                tape.backward(ahead, lo)
                tape.done_to = ahead
                _wait_mark(dev, mark)
                del tape.side_keep[:nkeep]   # (what that stream had read up to the mark may be recycled now)
            else:
                if tape.used_side:
                    join_side(gout.device)
                tape.side_keep.clear()
        else:
            tape.backward()
            if tape.used_side:
                join_side(gout.device)          # weight gradients issued on the side stream
            tape.side_keep.clear()              # (what that stream read may be recycled now: the main stream is ordered behind it)
        need = ctx.needs_input_grad[4:]
        res = []
        for i, a in enumerate(ctx.acts):
            g = tape.grads.get(id(a)) if need[i] else None
            if g is None:
                res.append(None)
            elif ctx.in_cl[i]:
                res.append(g.to_tensor())
            else:
                res.append(g.to_nchw().reshape(ctx.in_shapes[i]))
        if ctx.slice is not None and ctx.slice[0] > 0:
            # an inner or last segment: its own parameters' gradients leave now -- stock DistributedDataParallel's reducer sees them
            # (and sends its buckets) while the segments in front of this one still run their backward; the gradient of the token
            # that stands for the previous segment is a constant
            res = [_token_grad(gout.device)]
            for j, p in enumerate(ctx.params):
                sunk = id(p) in tape.sunk
                res.append(tape.param_grads.pop(id(p), None) if (need[ctx.n_in + j] and not sunk) else None)
                tape.delivered.add(id(p))
            if tape.gscale != 1.0:                 # half-precision mode: the loss scale leaves the parameter gradients here
                scale_tensors_([g for g in res[1:] if g is not None], 1.0 / tape.gscale)
            ctx.acts = ctx.params = ctx.out_act = ctx.holder = None
            return (None, None, None, None, *res)
        n_act = len(res)
        for j, p in enumerate(ctx.params):
            # gradients the engine wrote into a GradSync bucket are handed over by GradSync.finish(), not by autograd
            sunk = id(p) in tape.sunk
            res.append(tape.param_grads.get(id(p)) if (need[ctx.n_in + j] and not sunk) else None)
        if tape.gscale != 1.0:
            # half-precision mode: the loss scale leaves the parameter gradients here (gradients in a GradSync bucket lost it before
            # their hand-over: engine.Tape.backward); an input gradient carries it too
            scale_tensors_([g for g in res[n_act:] if g is not None] + [g.contiguous() for g in res[:n_act] if g is not None and g.is_contiguous()],
                           1.0 / tape.gscale)
            if any(g is not None and not g.is_contiguous() for g in res[:n_act]):
                raise RuntimeError("hyperpri_amd: precision 'f16' cannot return a channels-last input gradient (whole networks only)")
        tape.sunk.clear()
        tape.grads.clear()
        tape.param_grads.clear()
        ctx.acts = ctx.params = ctx.out_act = ctx.holder = None
        return (None, None, None, None, *res)


# ---- the same node as a PyTorch-ROCm custom operator (torch.library; BASELINE.json north_star: "registered as PyTorch-ROCm custom
# ops through a thin C-ABI extension") ---------------------------------------------------------------------------------------------
# One operator per reference module, all with the schema
#     hyperpri::<name>(Tensor[] inputs, Tensor[] params, int program, bool grad_mode, int input_planes) -> Tensor
# registered for the CUDA (= ROCm) dispatch key only -- there is no CPU kernel to fall back to -- with its backward registered
# through ``register_autograd``.  ``program`` is a handle into the table of tape programs of the calling module (the op graph the
# reference module stands for, closed over the module for its BatchNorm buffers, which a training-mode forward updates in place as
# nn.BatchNorm does); the kernels behind it are reached through the C ABI (include/hyperpri_hip.h) exactly as from _HipFn.
# The BatchNorm buffers a training-mode forward updates are NOT declared as mutated arguments: torch.library refuses an autograd formula
# on a mutating operator ("Cannot register autograd formula for non-functional operator": tried in round 5), and the functional
# alternative -- the operator returning new buffer values for Python to copy back -- would put 81 small copy kernels into every step.
# The operators are therefore correct in eager mode (what the reference's callers run) and not meant for torch.compile / export.
# HPRI_DISPATCHER=0 routes the modules through the plain autograd.Function instead (same tape, same kernels, same results).
OP_NAMES = ("unet", "cubenet", "cubenet_stem", "cubenet_up4", "spectral_unet", "double_conv", "down", "up", "out_conv", "run_program", "segment")
USE_DISPATCHER = os.environ.get("HPRI_DISPATCHER", "1") != "0"
_PROGRAMS: dict = {}
_HANDLES = itertools.count(1)
_TLS = threading.local()


class _Ctx:      # what _HipFn keeps on its ctx, for the operator path
    pass


def _op_forward(inputs: List[torch.Tensor], params: List[torch.Tensor], program: int, grad_mode: bool, input_planes: int) -> torch.Tensor:
    prog, kind = _PROGRAMS[program]
    st = _Ctx()
    st.handle = program
    st.needs_input_grad = (False, False, False, False, *[t.requires_grad for t in inputs], *[p.requires_grad for p in params])
    with torch.cuda.device(inputs[0].device):
        res = _HipFn._forward(st, prog, len(inputs), tuple(params), (grad_mode, int(input_planes), kind), *inputs, *params)
    # picked up by _op_setup_context, which the autograd kernel calls right after this forward (nothing to keep when nothing was recorded)
    _TLS.last = st if getattr(st, "tape", None) is not None else None
    return res


def _op_setup_context(ctx, inputs, output):
    ctx.st = getattr(_TLS, "last", None)
    _TLS.last = None
    if ctx.st is not None and ctx.st.handle != inputs[2]:       # (the state left by ANOTHER call of the operator: never attach it)
        ctx.st = None
        raise RuntimeError("hyperpri_amd: internal error: the tape handed to setup_context belongs to another operator call")


def _op_backward(ctx, gout):
    st = ctx.st
    if st is None or getattr(st, "tape", None) is None:
        raise RuntimeError("hyperpri_amd: backward called twice (retain_graph is not supported)")
    n_in = st.n_in
    with torch.cuda.device(gout.device):
        res = _HipFn._backward(st, gout)[4:]
    return list(res[:n_in]), list(res[n_in:]), None, None, None


def _register_ops():
    ops = {}
    for name in OP_NAMES:
        op = torch.library.custom_op(f"hyperpri::{name}", _op_forward_named(name), mutates_args=(), device_types="cuda")
        op.register_autograd(_op_backward, setup_context=_op_setup_context)
        ops[name] = op
    return ops


def _op_forward_named(name):
    def fwd(inputs: List[torch.Tensor], params: List[torch.Tensor], program: int, grad_mode: bool, input_planes: int) -> torch.Tensor:
        return _op_forward(inputs, params, program, grad_mode, input_planes)
    fwd.__name__ = name
    return fwd


OPS = _register_ops()


def run(program: Callable, inputs: Sequence[torch.Tensor], params: Sequence[torch.Tensor], input_planes: int = 0,
        name: str = "run_program", lib_kind: Optional[str] = None) -> torch.Tensor:
    """Run ``program`` as one autograd node.  ``params`` are the nn.Parameters the program reads (the
    program closes over the owning module; they are listed here so autograd routes their gradients).
    ``input_planes`` > 0: the layout pass of an NCHW input also writes that many bf16 planes (bf16 plane mode).
    ``name``: the custom operator the call goes through (``torch.ops.hyperpri.<name>``)."""
    dev = inputs[0].device
    for p in params:
        _require_cuda(p, "module parameter")
        if p.device != dev:
            raise RuntimeError(f"hyperpri_amd: parameter on {p.device} but input on {dev}")
        if not p.is_contiguous():
            raise RuntimeError("hyperpri_amd: parameters must be contiguous")
    for t in inputs:
        _require_cuda(t, "input tensor")
    if USE_DISPATCHER:
        h = next(_HANDLES)
        _PROGRAMS[h] = (program, lib_kind)
        try:
            return getattr(torch.ops.hyperpri, name)(list(inputs), list(params), h, torch.is_grad_enabled(), int(input_planes))
        finally:
            _PROGRAMS.pop(h, None)
    return _HipFn.apply(program, len(inputs), tuple(params), (torch.is_grad_enabled(), int(input_planes), lib_kind), *inputs, *params)
