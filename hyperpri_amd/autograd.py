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
from typing import Callable, List, Sequence

import torch

from .engine import Act, Tape, _require_cuda, join_side


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


class _HipFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, program: Callable, n_in: int, params: Sequence[torch.Tensor], grad_on, *tensors: torch.Tensor):
        with torch.cuda.device(tensors[0].device):      # launches and current_stream() follow the data's device
            return _HipFn._forward(ctx, program, n_in, params, grad_on, *tensors)

    @staticmethod
    def _forward(ctx, program: Callable, n_in: int, params: Sequence[torch.Tensor], grad_on, *tensors: torch.Tensor):
        grad_on, input_planes = grad_on          # (caller's grad mode, bf16 planes wanted for the inputs)
        inputs = tensors[:n_in]
        need = list(ctx.needs_input_grad[4:])
        # needs_input_grad is True for trainable parameters even under torch.no_grad() / inference_mode(), and
        # is_grad_enabled() is always False inside Function.forward: the caller's grad mode is passed in by run()
        record = grad_on and any(need)
        tape = Tape(record)
        acts: List[Act] = []
        for t in inputs:
            _require_cuda(t, "input tensor")
            acts.append(Act.from_tensor(_as4d(t), input_planes))
        out = program(tape, acts, need[:n_in])
        if isinstance(out, Act):
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
        tape: Tape = ctx.tape
        if tape is None:
            raise RuntimeError("hyperpri_amd: backward called twice (retain_graph is not supported)")
        ctx.tape = None
        if ctx.out_act is not None:
            tape.grads[id(ctx.out_act)] = Act.from_tensor(gout)
        else:
            ctx.holder["g"] = gout
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
        for j, p in enumerate(ctx.params):
            # gradients the engine wrote into a GradSync bucket are handed over by GradSync.finish(), not by autograd
            sunk = id(p) in tape.sunk
            res.append(tape.param_grads.get(id(p)) if (need[ctx.n_in + j] and not sunk) else None)
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
# HPRI_DISPATCHER=0 routes the modules through the plain autograd.Function instead (same tape, same kernels, same results).
OP_NAMES = ("unet", "cubenet", "cubenet_stem", "cubenet_up4", "spectral_unet", "double_conv", "down", "up", "out_conv", "run_program")
USE_DISPATCHER = os.environ.get("HPRI_DISPATCHER", "1") != "0"
_PROGRAMS: dict = {}
_HANDLES = itertools.count(1)
_TLS = threading.local()


class _Ctx:      # what _HipFn keeps on its ctx, for the operator path
    pass


def _op_forward(inputs: List[torch.Tensor], params: List[torch.Tensor], program: int, grad_mode: bool, input_planes: int) -> torch.Tensor:
    prog = _PROGRAMS[program]
    st = _Ctx()
    st.needs_input_grad = (False, False, False, False, *[t.requires_grad for t in inputs], *[p.requires_grad for p in params])
    with torch.cuda.device(inputs[0].device):
        res = _HipFn._forward(st, prog, len(inputs), tuple(params), (grad_mode, int(input_planes)), *inputs, *params)
    # picked up by _op_setup_context, which the autograd kernel calls right after this forward (nothing to keep when nothing was recorded)
    _TLS.last = st if getattr(st, "tape", None) is not None else None
    return res


def _op_setup_context(ctx, inputs, output):
    ctx.st = getattr(_TLS, "last", None)
    _TLS.last = None


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
        name: str = "run_program") -> torch.Tensor:
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
        _PROGRAMS[h] = program
        try:
            return getattr(torch.ops.hyperpri, name)(list(inputs), list(params), h, torch.is_grad_enabled(), int(input_planes))
        finally:
            _PROGRAMS.pop(h, None)
    return _HipFn.apply(program, len(inputs), tuple(params), (torch.is_grad_enabled(), int(input_planes)), *inputs, *params)
