"""Per-image data parallelism: bucketed gradient all-reduce overlapped with backward.

The reference trains with Lightning ``strategy="ddp"`` (PLTrainer.py:434-442): one process per GPU,
per-rank batch 2, un-synced BatchNorm, gradients averaged by an all-reduce.  The modules here are
ordinary ``nn.Module``s, so stock ``DistributedDataParallel`` works unchanged; ``GradSync`` is the
MI355X-shaped equivalent used by bench.py: parameters are grouped into a few large flat buckets in
reverse registration order (= the order backward produces them), and as soon as the last gradient of
a bucket lands its all-reduce is issued asynchronously on RCCL's stream (backend "nccl" is RCCL on
ROCm; xGMI is point-to-point, so few large messages beat many small ones).  ``finish()`` waits,
averages and hands the reduced values back as ``param.grad``.
Works with any backend (``gloo`` on CPU for tests).
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class _Bucket:
    __slots__ = ("params", "offsets", "flat", "pending", "work")

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.offsets = []
        n = 0
        for p in params:
            self.offsets.append(n)
            n += p.numel()
        self.flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        self.pending = len(params)
        self.work = None


class GradSync:
    def __init__(self, module: torch.nn.Module, bucket_mb: float = 48.0, process_group=None,
                 broadcast_buffers: bool = False, force: bool = False):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.collective = self.world > 1 or (force and dist.is_initialized())   # force: rehearse with one rank
        params = [p for p in module.parameters() if p.requires_grad]
        cap = int(bucket_mb * (1 << 20))
        self.buckets: List[_Bucket] = []
        cur: List[torch.nn.Parameter] = []
        size = 0
        for p in reversed(params):           # backward produces gradients in reverse registration order
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= cap:
                self.buckets.append(_Bucket(cur))
                cur, size = [], 0
        if cur:
            self.buckets.append(_Bucket(cur))
        self._where = {}
        self._hooks = []
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[id(p)] = (b, i)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.module = module
        self.broadcast_buffers = broadcast_buffers

    # torch DDP broadcasts BN buffers from rank 0 every forward (broadcast_buffers=True default);
    # call this before eval/checkpoint to mimic it
    def sync_buffers(self) -> None:
        if self.world > 1:
            for b in self.module.buffers():
                dist.broadcast(b, 0, group=self.group)

    def _on_grad(self, p: torch.Tensor) -> None:
        b, i = self._where[id(p)]
        n = p.numel()
        b.flat[b.offsets[i]:b.offsets[i] + n].copy_(p.grad.reshape(-1))
        b.pending -= 1
        if b.pending == 0 and self.collective:
            b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self) -> None:
        """Wait for the outstanding all-reduces and install the averaged gradients."""
        for b in self.buckets:
            if b.pending != 0:
                # parameters that got no gradient this step contribute zeros
                for i, p in enumerate(b.params):
                    if p.grad is None:
                        b.flat[b.offsets[i]:b.offsets[i] + p.numel()].zero_()
                if self.collective and b.work is None:
                    b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if b.work is not None:
                b.work.wait()
                b.work = None
            if self.world > 1:
                b.flat.div_(self.world)
            for i, p in enumerate(b.params):
                p.grad = b.flat[b.offsets[i]:b.offsets[i] + p.numel()].view_as(p)
            b.pending = len(b.params)

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
