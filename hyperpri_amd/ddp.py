"""Per-image data parallelism: bucketed gradient all-reduce overlapped with backward.

The reference trains with Lightning ``strategy="ddp"`` (PLTrainer.py:434-442): one process per GPU,
per-rank batch 2, un-synced BatchNorm, gradients averaged by an all-reduce.  The modules here are
ordinary ``nn.Module``s, so stock ``DistributedDataParallel`` works unchanged; ``GradSync`` is the
MI355X-shaped equivalent used by bench.py:

* parameters are grouped into a few large flat buckets in reverse registration order (= the order
  backward produces them; xGMI is point-to-point, so few large messages beat many small ones);
* the buckets ARE the gradient storage: ``GradSync`` registers itself as the engine's gradient sink, so the
  weight-gradient kernels write straight into their slice of a bucket (no per-tensor copy), and the tape tells
  the sink the moment a parameter's gradient is final -- also when the whole network runs as ONE autograd node;
* as soon as the last gradient of a bucket has landed its all-reduce is issued asynchronously on RCCL's
  stream (backend "nccl" is RCCL on ROCm), overlapping the rest of backward; ``finish()`` waits, averages
  and installs the bucket views as ``param.grad``.

Gradients that reach a parameter through plain autograd (a foreign module, a CPU test) arrive through a
post-accumulate-grad hook and are copied into the bucket instead.  Works with any backend (``gloo`` on CPU).
"""
from __future__ import annotations

import contextlib
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from . import engine


class _Bucket:
    __slots__ = ("params", "offsets", "flat", "pending", "work", "landed", "t_issue")

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.offsets = []
        n = 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4          # 16-byte aligned slices: the HIP kernels store float4
        self.flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        self.pending = len(params)
        self.work = None
        self.landed: set = set()
        self.t_issue = None

    def view(self, i: int) -> torch.Tensor:
        p = self.params[i]
        return self.flat[self.offsets[i]:self.offsets[i] + p.numel()].view_as(p)


class GradSync:
    def __init__(self, module: torch.nn.Module, bucket_mb: Optional[float] = None, process_group=None,
                 broadcast_buffers: bool = False, force: bool = False, tail_mb: Optional[float] = None, reserve_cus: int = 0):
        if bucket_mb is None:
            bucket_mb = 24.0           # few large messages: xGMI is point-to-point (7 links x ~153 GB/s), a ring is per-link bound
        if tail_mb is None:
            tail_mb = 2.0              # the last bucket's all-reduce is the only one that cannot overlap with backward
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.collective = self.world > 1 or (force and dist.is_initialized())   # force: rehearse with one rank
        # RCCL averages inside the collective (ncclAvg); gloo (the CPU tests) sums and finish() divides
        self._avg = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
        params = [p for p in module.parameters() if p.requires_grad]
        ready = list(reversed(params))       # backward produces gradients in reverse registration order
        nbytes = [p.numel() * p.element_size() for p in ready]
        # The all-reduce of the LAST bucket cannot overlap with anything: keep it small.  The gradients that land last (the
        # first layers: a few hundred KB each in the U-Nets) form a tail bucket of at most tail_mb; the bucket before it is
        # issued while those layers' backward -- the largest feature maps, several ms -- still runs.
        cap = int(bucket_mb * (1 << 20))
        tail_cap = min(int(tail_mb * (1 << 20)), cap)
        ntail, tsize = 0, 0
        while ntail < len(ready) - 1 and tsize + nbytes[len(ready) - 1 - ntail] <= tail_cap:
            tsize += nbytes[len(ready) - 1 - ntail]
            ntail += 1
        self.buckets: List[_Bucket] = []
        cur: List[torch.nn.Parameter] = []
        size = 0
        for p, nb in zip(ready[:len(ready) - ntail], nbytes):
            cur.append(p)
            size += nb
            if size >= cap:
                self.buckets.append(_Bucket(cur))
                cur, size = [], 0
        if cur:
            self.buckets.append(_Bucket(cur))
        if ntail:
            self.buckets.append(_Bucket(ready[len(ready) - ntail:]))
        self._where: Dict[int, Tuple[_Bucket, int]] = {}
        self._hooks = []
        for b in self.buckets:
            for i, p in enumerate(b.params):
                self._where[id(p)] = (b, i)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.module = module
        self.broadcast_buffers = broadcast_buffers
        # torch DDP's default (broadcast_buffers=True): rank 0's buffers (BN running statistics) replace everyone's at the start of
        # every forward.  A pre-hook on the TOP-LEVEL module: the fused tape only steps aside for hooks on children.
        self._buf_hook = None
        if broadcast_buffers and self.world > 1:
            # broadcast_buffers=True  : every forward through the module, whatever its mode -- torch DDP's behaviour (a validation
            #                           forward on ALL ranks then reads rank 0's running statistics, as under Lightning DDP);
            # broadcast_buffers="train": training-mode forwards only -- for loops that validate on a subset of the ranks (rank-0-only
            #                           validation), where an eval forward must not enter a collective the other ranks never
            #                           join; call sync_buffers() on all ranks before such a validation to get DDP's values.
            if broadcast_buffers == "train":
                self._buf_hook = module.register_forward_pre_hook(lambda m, inp: self.sync_buffers() if m.training else None)
            else:
                self._buf_hook = module.register_forward_pre_hook(lambda m, inp: self.sync_buffers())
        if self.collective and engine.SIDE_STREAM and not engine.SIDE_STREAM_WITH_SINK and not GradSync._warned_queues:
            GradSync._warned_queues = True
            import sys
            print("hyperpri_amd.GradSync: GPU_MAX_HW_QUEUES could not be raised to 8 any more (the HIP runtime was initialised "
                  "before hyperpri_amd was imported); weight gradients stay on the main stream under the gradient sink "
                  "(~3 % slower backward).  Export GPU_MAX_HW_QUEUES=8 before the process starts.", file=sys.stderr, flush=True)
        # reserve_cus > 0: the fp32 Winograd weight gradient (one 104 KB workgroup per CU, grids of exact multiples of the CU count)
        # plans its pixel splits for that many fewer CUs, so that a collective's channels holding a few CUs do not push its last
        # workgroup into a second wave (2 x the launch: profiles/r05_hog_kernels_fp32.json).  Off by default: it changes the split,
        # i.e. the (still deterministic) summation order of those weight gradients, against a run without it.
        self.reserve_cus = int(reserve_cus)
        if self.reserve_cus > 0:
            engine.set_plan_option("wgrad_cu_reserve", self.reserve_cus)
        self._accumulating = False      # inside no_sync(): gradients add up locally, no collective
        self._micro = 0                 # backward passes since the last finish()
        self._direct: set = set()       # ids of parameters whose gradient the engine wrote into the bucket this step
        self._events = None             # (per-bucket issue events, end-of-finish event) of the last step
        # who issued the all-reduces since construction: "tape" = from inside the engine's backward tape (ready()), "hook" = from a
        # post-accumulate-grad hook (plain autograd), "finish" = left over for finish() (no overlap): tests and the bench line read it
        self.issued = {"tape": 0, "hook": 0, "finish": 0}
        self._issuer = "finish"
        engine.set_grad_sink(self)

    _warned_queues = False

    # torch DDP broadcasts BN buffers from rank 0 every forward (broadcast_buffers=True default): ONE coalesced broadcast there,
    # and one here -- the float buffers (running_mean / running_var) travel as one flat tensor per dtype, the integer ones
    # (num_batches_tracked) as another: 2 collectives per call instead of 54 for CubeNET.  Call it yourself before an
    # eval / checkpoint on all ranks to mimic DDP when broadcast_buffers is off.
    def sync_buffers(self) -> None:
        if self.world <= 1:
            return
        by_type: Dict[Tuple[torch.dtype, torch.device], List[torch.Tensor]] = {}
        for b in self.module.buffers():
            by_type.setdefault((b.dtype, b.device), []).append(b)
        for bufs in by_type.values():
            if len(bufs) == 1:
                dist.broadcast(bufs[0], 0, group=self.group)
                continue
            flat = torch.cat([b.reshape(-1) for b in bufs])
            dist.broadcast(flat, 0, group=self.group)
            off, views = 0, []
            for b in bufs:
                views.append(flat[off:off + b.numel()].view_as(b))
                off += b.numel()
            with torch.no_grad():
                torch._foreach_copy_(bufs, views)       # one multi-tensor launch back into the modules' buffers
        if any(b.is_cuda for b in self.module.buffers()):
            engine.bump_bn_epoch()          # folded eval packs depend on the running statistics

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside this context add into the buckets without communicating;
        the first backward after it (followed by ``finish()``) reduces the sum -- torch DDP's ``no_sync`` semantics."""
        old = self._accumulating
        self._accumulating = True
        try:
            yield
        finally:
            self._accumulating = old

    # ---- engine gradient sink -------------------------------------------------------------------------------------
    def slot(self, p: torch.Tensor):
        """Storage for p's gradient inside its bucket and whether the kernel must add to what is there."""
        w = self._where.get(id(p))
        if w is None:
            return None
        b, i = w
        self._guard(b, i)
        self._direct.add(id(p))
        return b.view(i), (self._micro > 0)

    def ready(self, p: torch.Tensor) -> None:
        b, i = self._where[id(p)]
        self._issuer = "tape"
        try:
            self._landed(b, i)
        finally:
            self._issuer = "finish"

    def completes_bucket(self, p: torch.Tensor) -> bool:
        """Will ``ready(p)`` issue a bucket's all-reduce?  (The engine joins its second stream first in that case.)"""
        b, i = self._where[id(p)]
        return b.pending == 1 and i not in b.landed and self.collective and not self._accumulating

    # ---- plain autograd path ----------------------------------------------------------------------------------------
    def _on_grad(self, p: torch.Tensor) -> None:
        if id(p) in self._direct:
            return
        b, i = self._where[id(p)]
        self._guard(b, i)
        v = b.view(i)
        if p.grad.data_ptr() != v.data_ptr():     # (p.grad may BE the view: installed by the previous finish())
            v.copy_(p.grad)                       # with micro-batches autograd has already summed them into p.grad
        self._issuer = "hook"
        try:
            self._landed(b, i)
        finally:
            self._issuer = "finish"

    def _guard(self, b: _Bucket, i: int) -> None:
        if b.work is not None or i in b.landed:
            raise RuntimeError(
                "hyperpri_amd.GradSync: a second backward reached a bucket whose all-reduce is already issued. "
                "Call finish() after every backward, or wrap all but the last micro-batch in `with sync.no_sync():`.")

    def _landed(self, b: _Bucket, i: int) -> None:
        b.landed.add(i)
        b.pending -= 1
        if b.pending == 0 and self.collective and not self._accumulating:
            self._issue(b)

    def _issue(self, b: _Bucket) -> None:
        self.issued[self._issuer] += 1
        if b.flat.is_cuda:
            b.t_issue = torch.cuda.Event(enable_timing=True)
            b.t_issue.record()
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.group, async_op=True)

    def end_micro_batch(self) -> None:
        """Close an accumulation micro-batch (called by the user after each backward inside ``no_sync``)."""
        for b in self.buckets:
            b.pending = len(b.params)
            b.landed.clear()
        self._direct.clear()
        self._micro += 1

    def finish(self) -> None:
        """Wait for the outstanding all-reduces and install the averaged gradients."""
        if self._accumulating:
            self.end_micro_batch()
            return
        for b in self.buckets:
            if b.pending != 0:
                # parameters that got no gradient this step contribute zeros (unless earlier micro-batches did)
                if self._micro == 0:
                    for i, p in enumerate(b.params):
                        if i not in b.landed:
                            b.view(i).zero_()
                if self.collective and b.work is None:
                    self._issue(b)
            if b.work is not None:
                b.work.wait()
                b.work = None
            if self.world > 1 and not self._avg:
                b.flat.div_(self.world)
            for i, p in enumerate(b.params):
                p.grad = b.view(i)
            b.pending = len(b.params)
            b.landed.clear()
        firsts = [b.t_issue for b in self.buckets if b.t_issue is not None]
        if firsts:
            t_end = torch.cuda.Event(enable_timing=True)
            t_end.record()
            self._events = (firsts, t_end)
        for b in self.buckets:
            b.t_issue = None
        self._direct.clear()
        self._micro = 0

    def overlap_ms(self) -> Optional[dict]:
        """After a device sync: per-bucket time from the issue of its all-reduce to the return of ``finish()``
        (stream time on the compute stream; the first entry is the window in which communication ran beside backward)."""
        ev = self._events
        if not ev:
            return None
        firsts, t_end = ev
        return {"buckets": len(self.buckets), "bucket_mb": [round(b.flat.numel() * 4 / 2 ** 20, 1) for b in self.buckets],
                "issue_to_finish_ms": [round(e.elapsed_time(t_end), 3) for e in firsts]}

    def remove(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
        if self._buf_hook is not None:
            self._buf_hook.remove()
            self._buf_hook = None
        if engine._GRAD_SINK is self:
            engine.set_grad_sink(None)
        if self.reserve_cus > 0:
            engine.set_plan_option("wgrad_cu_reserve", 0)
