"""The caller-side tail of a step on the GPU: loss, segmentation metrics, PR curve, optimizer, checkpoints.

SURVEY.md 8f rank 2-4.  Everything here mirrors what ``src/PLTrainer.py`` does around the network call --
names and semantics follow ``RootLightningModel`` (PLTrainer.py:34-183) and its evaluation helpers
(:270-330, :525-562) -- but none of it needs Lightning or torchmetrics, and every reduction runs in the HIP
library (``csrc/step.hip``) without a host synchronisation until a value is actually read.

    crit = BCEWithLogitsLoss()                       # drop-in for nn.BCEWithLogitsLoss() (params_HyperPRI.py:60)
    opt  = FusedAdam(net.parameters(), lr=1e-3)      # drop-in for optim.Adam (PLTrainer.py:171-174)
    model = SegmentationModel(net, crit, optimizer="Adam", lr=1e-3)
    loss = model.training_step({"image": x, "mask": m})

There is no CPU fallback: tensors must be fp32 on a ROCm device.
"""
from __future__ import annotations

import ctypes
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Tuple

import torch
from torch import nn

from . import _lib
from . import engine as _engine
from .engine import _p, _require_cuda, _stream, bump_param_epoch


def _flat(t: torch.Tensor, what: str) -> torch.Tensor:
    _require_cuda(t, what)
    return t if t.is_contiguous() else t.contiguous()


# ---------------------------------------------------------------------------------------------------
# nn.BCEWithLogitsLoss() (mean)
# ---------------------------------------------------------------------------------------------------
class _BCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred: torch.Tensor, target: torch.Tensor):
        _require_cuda(pred, "BCEWithLogitsLoss input")
        with torch.cuda.device(pred.device):
            x, y = _flat(pred, "BCEWithLogitsLoss input"), _flat(target, "BCEWithLogitsLoss target")
            if x.shape != y.shape:
                raise ValueError(f"Target size ({tuple(y.shape)}) must be the same as input size ({tuple(x.shape)})")
            n = x.numel()
            nws = _lib.load().hpri_bce_workspace_doubles(n)
            ws = torch.empty(nws, dtype=torch.float64, device=x.device)
            loss = torch.empty((), dtype=torch.float32, device=x.device)
            _lib.call("hpri_bce_logits_fwd", _p(x), _p(y), n, _p(loss), _p(ws), nws, _stream())
            ctx.save_for_backward(x, y)
            ctx.shape = pred.shape
        return loss

    @staticmethod
    def backward(ctx, gout: torch.Tensor):
        x, y = ctx.saved_tensors
        with torch.cuda.device(x.device):
            g = gout.contiguous().to(torch.float32)
            dx = torch.empty_like(x)
            _lib.call("hpri_bce_logits_bwd", _p(x), _p(y), x.numel(), _p(g), _p(dx), _stream())
        return dx.view(ctx.shape), None


class _FusedLossFn(torch.autograd.Function):
    """The autograd edge between the logits and a loss the head's kernel has already computed (engine.out_conv with a
    pending target).  Backward hands the scalar gradient to the head's backward kernels through the tape's holder and
    returns an all-zero stride-0 marker instead of a gradient tensor: no pass over the logits happens here."""

    @staticmethod
    def forward(ctx, logits: torch.Tensor, slot):
        ctx.holder, ctx.shape = slot.holder, logits.shape
        return slot.loss

    @staticmethod
    def backward(ctx, gout: torch.Tensor):
        with torch.cuda.device(gout.device):
            g = gout.contiguous().to(torch.float32)
            marker = torch.zeros(1, dtype=torch.float32, device=g.device).expand(ctx.shape)
        ctx.holder["bce_g"], ctx.holder["bce_marker"] = g, marker
        return marker, None


def forward_loss(network: nn.Module, image: torch.Tensor, target: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """``pred = network(image); loss = nn.BCEWithLogitsLoss()(pred, target)`` (PLTrainer.py:85-86) as one call, so that the
    loss is computed inside the network's last layer (forward: the 1x1 head also leaves the BCE partial sums; backward:
    the head's gradient kernels form (sigmoid(pred) - target)/n themselves).  Same values as the two-call form up to
    fp32 summation order inside the head's weight gradient; falls back to the two-call form when the network's head does
    not take the offer (hooks, ``fused_tape = False``, no gradient recording).

    What differs from the two-call form, by design: the autograd edge from the loss to ``pred`` carries an all-zero stride-0
    marker, not ``(sigmoid(pred) - target) / n`` -- the head's backward kernels form that product themselves.  So
    ``torch.autograd.grad(loss, pred)``, ``pred.retain_grad()`` and tensor hooks on ``pred`` see zeros for the loss's share
    (parameter and input gradients are complete and bit-identical to the two-call form).  Code that inspects dLoss/dLogits should
    call ``criterion(network(image), target)`` instead.  The logits and the target are re-read in backward: an in-place write to
    either in between raises, as it would for tensors saved by ``nn.BCEWithLogitsLoss``."""
    _require_cuda(image, "input tensor")
    with torch.cuda.device(image.device):
        tgt = _flat(target if target.dtype == torch.float32 else target.to(torch.float32), "BCEWithLogitsLoss target")
    with _engine.pending_bce(tgt) as slot:
        pred = network(image)
    if isinstance(pred, tuple):                      # analyze=True networks return (pred, features)
        pred = pred[0]
    if pred.shape != target.shape:
        # nn.BCEWithLogitsLoss (and _BCEFn) refuse e.g. an (N,H,W) mask against (N,1,H,W) logits; so does the fused form, whose
        # kernels only ever compared element counts
        if slot.holder is not None:
            slot.holder.clear()
        raise ValueError(f"Target size ({tuple(target.shape)}) must be the same as input size ({tuple(pred.shape)})")
    if slot.used and slot.holder is not None and pred.requires_grad and pred.numel() == tgt.numel():
        return pred, _FusedLossFn.apply(pred, slot)
    if pred.shape != target.shape:
        raise ValueError(f"Target size ({tuple(target.shape)}) must be the same as input size ({tuple(pred.shape)})")
    return pred, _BCEFn.apply(pred, target)


class BCEWithLogitsLoss(nn.Module):
    """``nn.BCEWithLogitsLoss()`` with mean reduction (params_HyperPRI.py:60,223): one fused pass forward
    (fp64 partial sums in a fixed order -> bit-reproducible), one pass backward."""

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:   # noqa: A002 (torch's names)
        return _BCEFn.apply(input, target)


# ---------------------------------------------------------------------------------------------------
# Accuracy / JaccardIndex / Dice from one confusion count (PLTrainer.py:62-68, 88-91)
# ---------------------------------------------------------------------------------------------------
class SegCounts:
    """TP/FP/FN/TN of ``sigmoid(pred) > threshold`` against ``mask.to(int32)``, accumulated on the device.

    ``compute()`` returns what the reference logs: pixel accuracy (``Accuracy(task='binary')``), positive-class
    Dice (``Dice(num_classes=2, ignore_index=0, zero_division=1e-12)``) and +IoU (``JaccardIndex('binary')``).
    """

    def __init__(self, threshold: float = 0.5, device=None):
        self.threshold = float(threshold)
        self.counts = None if device is None else torch.zeros(4, dtype=torch.int64, device=device)

    def reset(self) -> None:
        if self.counts is not None:
            self.counts.zero_()

    def update(self, pred: torch.Tensor, mask: torch.Tensor, is_logits: bool = True) -> None:
        _require_cuda(pred, "prediction")
        with torch.cuda.device(pred.device):
            x = _flat(pred.detach(), "prediction")
            y = _flat(mask if mask.dtype == torch.float32 else mask.to(torch.float32), "mask")
            if x.numel() != y.numel():
                raise ValueError("prediction and mask differ in size")
            if self.counts is None:
                self.counts = torch.zeros(4, dtype=torch.int64, device=x.device)
            _lib.call("hpri_seg_counts", _p(x), _p(y), x.numel(), self.threshold, int(is_logits), _p(self.counts), _stream())

    def compute(self) -> Dict[str, float]:
        tp, fp, fn, tn = (float(v) for v in self.counts.tolist())     # the one host synchronisation
        return metrics_from_counts(tp, fp, fn, tn)


class _StepCounts:
    """One (TP, FP, FN, TN) row per step, kept on the device ([capacity, 4] int64, no host sync until ``rows()``)."""

    def __init__(self, threshold: float, device, capacity: int = 256):
        self.threshold = float(threshold)
        self.buf = torch.zeros((capacity, 4), dtype=torch.int64, device=device)
        self.weights: List[int] = []
        self.n = 0

    def reset(self) -> None:
        self.buf.zero_()
        self.weights.clear()
        self.n = 0

    def update(self, pred: torch.Tensor, mask: torch.Tensor) -> None:
        _require_cuda(pred, "prediction")
        with torch.cuda.device(pred.device):
            if self.n == self.buf.shape[0]:
                grown = torch.zeros((2 * self.n, 4), dtype=torch.int64, device=self.buf.device)
                grown[:self.n].copy_(self.buf)
                self.buf = grown
            x = _flat(pred.detach(), "prediction")
            y = _flat(mask if mask.dtype == torch.float32 else mask.to(torch.float32), "mask")
            if x.numel() != y.numel():
                raise ValueError("prediction and mask differ in size")
            row = self.buf[self.n]
            _lib.call("hpri_seg_counts", _p(x), _p(y), x.numel(), self.threshold, 1, _p(row), _stream())
            self.weights.append(int(pred.shape[0]))
            self.n += 1

    def rows(self) -> Tuple[List[List[float]], List[int]]:
        return [[float(v) for v in r] for r in self.buf[:self.n].tolist()], list(self.weights)


def metrics_from_counts(tp: float, fp: float, fn: float, tn: float) -> Dict[str, float]:
    total = tp + fp + fn + tn
    return {
        "acc": (tp + tn) / total if total > 0 else 0.0,
        "dice": (2 * tp) / (2 * tp + fp + fn) if (2 * tp + fp + fn) > 0 else 1e-12,
        "pos_iou": tp / (tp + fp + fn) if (tp + fp + fn) > 0 else 0.0,
        "tp": tp, "fp": fp, "fn": fn, "tn": tn,
    }


# ---------------------------------------------------------------------------------------------------
# PrecisionRecallCurve('binary', thresholds=500) and the best-Dice threshold (PLTrainer.py:542-556)
# ---------------------------------------------------------------------------------------------------
class PRCurve:
    """Binned binary precision-recall curve (torchmetrics 1.2.0 semantics: ``pred >= threshold``, thresholds =
    ``torch.linspace(0, 1, T)``).  The per-pixel work is one histogram pass; the T confusion matrices are suffix
    sums of the two class histograms."""

    def __init__(self, thresholds: int = 500, device=None):
        self.T = int(thresholds)
        self.device = device
        self.thresholds = None
        self.hist = None

    def _ensure(self, device):
        if self.hist is None:
            self.thresholds = torch.linspace(0, 1, self.T, dtype=torch.float32).to(device)
            self.hist = torch.zeros(2 * (self.T + 1), dtype=torch.int64, device=device)

    def update(self, pred: torch.Tensor, target: torch.Tensor, is_logits: bool = False) -> None:
        _require_cuda(pred, "prediction")
        with torch.cuda.device(pred.device):
            x = _flat(pred.detach().reshape(-1), "prediction")
            y = _flat((target if target.dtype == torch.float32 else target.to(torch.float32)).reshape(-1), "target")
            if x.numel() != y.numel():
                raise ValueError("prediction and target differ in size")
            self._ensure(x.device)
            _lib.call("hpri_pr_curve_hist", _p(x), _p(y), x.numel(), _p(self.thresholds), self.T, int(is_logits),
                      _p(self.hist), _stream())

    def confusion(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """(tp, fp, fn, tn) per threshold, int64, on the host."""
        h = self.hist.cpu().view(2, self.T + 1)
        # bin b holds pixels with exactly b thresholds <= p, so pred >= t_k  <=>  b > k
        ge = torch.flip(torch.cumsum(torch.flip(h, dims=[1]), dim=1), dims=[1])[:, 1:]     # [class][k] = #{b > k}
        tot = h.sum(dim=1, keepdim=True)
        tp, fp = ge[1], ge[0]
        return tp, fp, tot[1] - tp, tot[0] - fp

    def compute(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(precision[T+1], recall[T+1], thresholds[T]) exactly as torchmetrics returns them."""
        tp, fp, fn, _ = self.confusion()
        tp, fp, fn = tp.to(torch.float32), fp.to(torch.float32), fn.to(torch.float32)

        def safe_div(a, b):
            b = torch.where(b == 0, torch.ones_like(b), b)
            return a / b

        precision = torch.cat([safe_div(tp, tp + fp), torch.ones(1)])
        recall = torch.cat([safe_div(tp, tp + fn), torch.zeros(1)])
        return precision, recall, self.thresholds.cpu()


def best_dice_threshold(precision: torch.Tensor, recall: torch.Tensor, thresholds: torch.Tensor):
    """The threshold pick of ``model_eval`` (PLTrainer.py:546-556): drop the top and bottom 1 % of the curve,
    Dice = 2PR/(P+R), arg-max, threshold rounded to 2 decimals.  Returns (threshold, precision, recall)."""
    crop = int(len(precision) // 100)
    p, r, t = precision[crop:-crop], recall[crop:-crop], thresholds[crop:-crop]
    dice = 2 * p * r / (p + r)
    i = int(torch.argmax(dice))
    return float(torch.round(t[i].to(torch.float), decimals=2)), float(p[i]), float(r[i])


def average_precision(pred: torch.Tensor, target: torch.Tensor) -> float:
    """``AveragePrecision(task='binary')`` of ``model_eval`` (PLTrainer.py:558-559, 650-651; torchmetrics 1.2.0 with
    ``thresholds=None``): AP = sum_n (R_n - R_{n-1}) P_n over the distinct prediction values in descending order --
    the same definition as ``sklearn.metrics.average_precision_score``.  Exact (no binning): one device sort and two
    cumulative sums with torch's own kernels (plumbing, not the hot path); ties share one threshold."""
    _require_cuda(pred, "prediction")
    p = pred.detach().reshape(-1).to(torch.float32)
    t = (target.reshape(-1).to(torch.int32) != 0)
    order = torch.argsort(p, descending=True, stable=True)
    ps, ts = p[order], t[order]
    tp = torch.cumsum(ts.to(torch.float64), 0)
    fp = torch.cumsum((~ts).to(torch.float64), 0)
    last = torch.ones_like(ps, dtype=torch.bool)            # last element of every run of equal predictions
    last[:-1] = ps[1:] != ps[:-1]
    tp, fp = tp[last], fp[last]
    npos = tp[-1]
    if float(npos) == 0.0:
        return float("nan")
    precision = tp / (tp + fp)
    recall = tp / npos
    prev = torch.cat([torch.zeros(1, dtype=torch.float64, device=recall.device), recall[:-1]])
    return float(((recall - prev) * precision).sum())


# ---------------------------------------------------------------------------------------------------
# optim.Adam / optim.SGD as one multi-tensor launch
# ---------------------------------------------------------------------------------------------------
def _ptr_array(ts: List[Optional[torch.Tensor]]):
    return (ctypes.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])


class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam(params, lr, betas, eps, weight_decay)`` (amsgrad/maximize off -- the reference uses the
    defaults, PLTrainer.py:171-174) with every parameter tensor of a group updated by ONE kernel launch.
    ``grad_scale``: optional device scalar multiplied into the gradients (e.g. 1/world_size)."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, grad_scale: Optional[torch.Tensor] = None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            for p in ps:
                _require_cuda(p, "parameter")
                if p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters and gradients must be contiguous fp32")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            by_step: Dict[int, List[torch.Tensor]] = OrderedDict()
            for p in ps:
                self.state[p]["step"] += 1
                by_step.setdefault(self.state[p]["step"], []).append(p)
            b1, b2 = group["betas"]
            for step, plist in by_step.items():
                with torch.cuda.device(plist[0].device):
                    n = (ctypes.c_longlong * len(plist))(*[p.numel() for p in plist])
                    _lib.call("hpri_adam_step", _ptr_array(plist), _ptr_array([p.grad for p in plist]),
                              _ptr_array([self.state[p]["exp_avg"] for p in plist]),
                              _ptr_array([self.state[p]["exp_avg_sq"] for p in plist]), n, len(plist),
                              float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                              int(step), _p(grad_scale), _stream())
        bump_param_epoch()      # parameters were written through raw pointers: packed-weight caches are stale
        return loss


class FusedSGD(torch.optim.Optimizer):
    """``torch.optim.SGD(params, lr, momentum, weight_decay)`` (dampening 0, no Nesterov; PLTrainer.py:176-180)."""

    def __init__(self, params, lr: float = 1e-3, momentum: float = 0.0, weight_decay: float = 0.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("FusedSGD: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None, grad_scale: Optional[torch.Tensor] = None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            mom = float(group["momentum"])
            fresh, old = [], []
            for p in ps:
                _require_cuda(p, "parameter")
                if p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or not p.is_contiguous():
                    raise RuntimeError("FusedSGD: parameters and gradients must be contiguous fp32")
                st = self.state[p]
                if mom != 0.0 and "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.preserve_format)
                    fresh.append(p)
                else:
                    old.append(p)
            for first, plist in ((1, fresh), (0, old)):
                if not plist:
                    continue
                with torch.cuda.device(plist[0].device):
                    n = (ctypes.c_longlong * len(plist))(*[p.numel() for p in plist])
                    bufs = _ptr_array([self.state[p]["momentum_buffer"] for p in plist]) if mom != 0.0 else None
                    _lib.call("hpri_sgd_step", _ptr_array(plist), _ptr_array([p.grad for p in plist]), bufs, n, len(plist),
                              float(group["lr"]), mom, float(group["weight_decay"]), first, _p(grad_scale), _stream())
        bump_param_epoch()
        return loss


# ---------------------------------------------------------------------------------------------------
# RootLightningModel without Lightning (PLTrainer.py:34-183)
# ---------------------------------------------------------------------------------------------------
class SegmentationModel(nn.Module):
    """The step logic of ``RootLightningModel``: same attribute names (``m_network``, ``f_criterion``,
    ``threshold``, ``predict_labels``), same step methods, metrics accumulated on the device and read by
    ``epoch_metrics`` (the ``on_epoch=True`` logging of the reference)."""

    def __init__(self, network: nn.Module, criterion: Optional[nn.Module] = None, optimizer: str = "Adam",
                 lr: float = 1e-3, weight_decay: float = 0.0, momentum: float = 0.9, threshold: float = 0.5):
        super().__init__()
        self.m_network = network
        self.f_criterion = criterion if criterion is not None else BCEWithLogitsLoss()
        self.p_optimizer, self.p_learn_rate, self.p_decay, self.p_momentum = optimizer, lr, weight_decay, momentum
        self.threshold = threshold
        self.predict_labels: List[torch.Tensor] = []
        self._counts: Dict[str, "_StepCounts"] = {}
        self._loss: Dict[str, List[torch.Tensor]] = {}

    # -- PLTrainer.py:166-183
    def configure_optimizers(self):
        name = self.p_optimizer.upper()
        if name == "ADAM":
            return FusedAdam(self.m_network.parameters(), lr=self.p_learn_rate, weight_decay=self.p_decay)
        if name == "SGD":
            return FusedSGD(self.m_network.parameters(), lr=self.p_learn_rate, momentum=self.p_momentum,
                            weight_decay=self.p_decay)
        raise RuntimeError(f"Optimizer {self.p_optimizer} not supported")

    def _forward(self, image: torch.Tensor) -> torch.Tensor:
        if getattr(self.m_network, "analyze", False):
            pred, _ = self.m_network(image)
            return pred
        return self.m_network(image)

    def _step(self, stage: str, batch, threshold: float) -> Tuple[torch.Tensor, torch.Tensor]:
        if type(self.f_criterion) is BCEWithLogitsLoss and torch.is_grad_enabled():
            pred, loss = forward_loss(self.m_network, batch["image"], batch["mask"])      # loss inside the head's kernels
        else:
            pred = self._forward(batch["image"])
            loss = self.f_criterion(pred, batch["mask"])
        c = self._counts.get(stage)
        if c is None or c.threshold != threshold:
            c = self._counts[stage] = _StepCounts(threshold, pred.device)
        c.update(pred, batch["mask"])
        self._loss.setdefault(stage, []).append(loss.detach())
        return pred, loss

    def training_step(self, batch, batch_idx: int = 0) -> torch.Tensor:       # PLTrainer.py:79-98
        return self._step("tr", batch, self.threshold)[1]

    def validation_step(self, batch, batch_idx: int = 0) -> None:             # PLTrainer.py:100-118 (threshold 0.5)
        self._step("val", batch, 0.5)

    def test_step(self, batch, batch_idx: int = 0) -> torch.Tensor:           # PLTrainer.py:120-140
        return self._step("test", batch, self.threshold)[0]

    def predict_step(self, batch, batch_idx: int = 0) -> torch.Tensor:        # PLTrainer.py:142-162
        self.predict_labels.append(batch["mask"].cpu())
        return self._forward(batch["image"]).cpu()

    def epoch_metrics(self, stage: str, reset: bool = True) -> Dict[str, float]:
        """{'<stage>_loss', '<stage>_acc', '<stage>_dice', '<stage>_pos_iou'} over the steps since the last reset, as
        the reference logs them: every step computes its own Accuracy / Dice / +IoU and ``self.log(..., on_epoch=True)``
        averages the per-step VALUES weighted by batch size (PLTrainer.py:88-96, 113-118) -- a mean of ratios, which is
        what ``ModelCheckpoint(monitor='val_dice')`` (PLTrainer.py:352) ranks checkpoints by.  The ratio of the epoch's
        summed counts (what one would report for the whole split) is returned beside it under ``<stage>_*_pooled``."""
        out: Dict[str, float] = {}
        if stage in self._loss and self._loss[stage]:
            out[f"{stage}_loss"] = float(torch.stack(self._loss[stage]).mean())
        if stage in self._counts and self._counts[stage].n:
            steps, weights = self._counts[stage].rows()      # the one host synchronisation
            wsum = float(sum(weights))
            per = [metrics_from_counts(*r) for r in steps]
            for k in ("acc", "dice", "pos_iou"):
                out[f"{stage}_{k}"] = sum(w * m[k] for w, m in zip(weights, per)) / wsum
            pooled = metrics_from_counts(*[sum(r[j] for r in steps) for j in range(4)])
            out.update({f"{stage}_acc_pooled": pooled["acc"], f"{stage}_dice_pooled": pooled["dice"],
                        f"{stage}_pos_iou_pooled": pooled["pos_iou"]})
        if reset:
            self._loss.pop(stage, None)
            if stage in self._counts:
                self._counts[stage].reset()
        return out


# ---------------------------------------------------------------------------------------------------
# checkpoint formats (PLTrainer.py:186-216, 270-330)
# ---------------------------------------------------------------------------------------------------
def network_state_dict(raw: dict) -> "OrderedDict[str, torch.Tensor]":
    """Whatever ``load_val_model`` accepts -> the network's own ``state_dict`` keys.

    * Lightning ``.ckpt`` (has ``'pytorch-lightning_version'``; weights under ``state_dict`` as ``m_network.<key>``),
    * raw ``best_wts.pt`` (plain keys, or ``module.<key>`` from ``nn.DataParallel``/DDP),
    * consolidated DeepSpeed ZeRO-2 (``_forward_module.m_network.<key>``; ``feat_ext`` entries dropped,
      PLTrainer.py:203-211).
    """
    sd = raw["state_dict"] if "pytorch-lightning_version" in raw else raw
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in sd.items():
        k = k.replace("_forward_module.m_network.", "")
        if "feat_ext" in k:
            continue
        if k.startswith("m_network."):
            k = k[len("m_network."):]
        elif "module." in k:
            k = k.replace("module.", "", 1)
        out[k] = v
    return out


def load_checkpoint(network: nn.Module, path: str) -> nn.Module:
    network.load_state_dict(network_state_dict(torch.load(path, map_location="cpu")))
    return network
