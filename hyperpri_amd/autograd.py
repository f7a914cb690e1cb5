"""torch.autograd bridge: one Function node per reference module call.

``run(program, inputs, params)`` executes ``program(tape, acts, need)`` -- a sequence of HIP ops from
``engine`` -- and returns an ordinary autograd-tracked tensor, as the reference's callers expect
(PLTrainer.py:85-88: BCEWithLogitsLoss(pred, mask), torch.sigmoid(pred.detach()), pred.cpu()).
Feature maps are returned as channels-last views (logical NCHW shape) so module-to-module hand-offs
never copy; logits are plain contiguous (N, n_classes, H, W).
Backward runs the tape in reverse on the autograd worker thread with the thread's current stream.
"""
from __future__ import annotations

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


def run(program: Callable, inputs: Sequence[torch.Tensor], params: Sequence[torch.Tensor], input_planes: int = 0) -> torch.Tensor:
    """Run ``program`` as one autograd node.  ``params`` are the nn.Parameters the program reads (the
    program closes over the owning module; they are listed here so autograd routes their gradients).
    ``input_planes`` > 0: the layout pass of an NCHW input also writes that many bf16 planes (bf16 plane mode)."""
    dev = inputs[0].device
    for p in params:
        _require_cuda(p, "module parameter")
        if p.device != dev:
            raise RuntimeError(f"hyperpri_amd: parameter on {p.device} but input on {dev}")
        if not p.is_contiguous():
            raise RuntimeError("hyperpri_amd: parameters must be contiguous")
    for t in inputs:
        _require_cuda(t, "input tensor")
    return _HipFn.apply(program, len(inputs), tuple(params), (torch.is_grad_enabled(), int(input_planes)), *inputs, *params)
