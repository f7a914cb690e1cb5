"""CPU ORACLE for the HyperPRI segmentation hot path -- TEST INFRASTRUCTURE ONLY.

This file restates, op by op, what the reference's ``src/Experiments/model_parts.py`` and
``src/Experiments/models.py`` compute, as plain functions over a ``state_dict``-style mapping
(name -> fp32 tensor) on the PyTorch **CPU** backend.  The reference itself is ~400 lines of
Python that delegates every arithmetic op to ``torch.nn`` (ATen/oneDNN on CPU; version left
unpinned by ``environment.yml:21``), so the restatement calls the same ATen CPU primitives
through ``torch.nn.functional`` -- there is no other arithmetic in the reference to restate.

Pinning: ``tests/golden/make_golden.py`` imports the real reference modules in the build
container, loads generator-defined weights into them and stores their outputs/gradients as
fixtures under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this file against every
one of those fixtures.  The reference ships no tests/golden vectors of its own (SURVEY.md
section 4), so those fixtures plus the two published parameter counts (``README.md:65``,
``test_models.ipynb:201``) are the pins.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (``hyperpri_amd``) never does.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

BN_EPS = 1e-5        # torch.nn.BatchNorm*d default, model_parts.py:23,26
BN_MOMENTUM = 0.1


# --------------------------------------------------------------------------------------------
# building blocks (model_parts.py)
# --------------------------------------------------------------------------------------------
def _bn(sd: SD, p: str, x: Tensor, train: bool) -> Tensor:
    """BatchNorm{1,2,3}d: batch statistics (biased var) in train mode and running-stat update with
    the unbiased var; running statistics in eval mode.  model_parts.py:23,26; models.py:113,172,178."""
    rm, rv = sd[p + ".running_mean"], sd[p + ".running_var"]
    y = F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], train, BN_MOMENTUM, BN_EPS)
    if train and (p + ".num_batches_tracked") in sd:
        sd[p + ".num_batches_tracked"] += 1
    return y


def double_conv(sd: SD, p: str, x: Tensor, train: bool) -> Tensor:
    """DoubleConv.forward, model_parts.py:14-31: (conv3x3 pad1 -> BN -> ReLU) x 2."""
    q = p + "double_conv."
    x = F.conv2d(x, sd[q + "0.weight"], sd[q + "0.bias"], padding=1)
    x = F.relu(_bn(sd, q + "1", x, train))
    x = F.conv2d(x, sd[q + "3.weight"], sd[q + "3.bias"], padding=1)
    x = F.relu(_bn(sd, q + "4", x, train))
    return x


def down(sd: SD, p: str, x: Tensor, train: bool) -> Tensor:
    """Down.forward, model_parts.py:34-45: MaxPool2d(2) (floor) -> DoubleConv."""
    return double_conv(sd, p + "maxpool_conv.1.", F.max_pool2d(x, 2), train)


def _pad_to(x1: Tensor, x2: Tensor) -> Tensor:
    """model_parts.py:73-80: zero-pad x1 to x2's H,W with left=floor(d/2), right=d-floor(d/2)."""
    dy = x2.shape[2] - x1.shape[2]
    dx = x2.shape[3] - x1.shape[3]
    return F.pad(x1, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])


def up(sd: SD, p: str, x1: Tensor, x2: Tensor, train: bool, bilinear: bool = False,
       use_attention: bool = False) -> Tensor:
    """Up.forward, model_parts.py:71-90."""
    if bilinear:
        x1 = F.interpolate(x1, scale_factor=2, mode="bilinear", align_corners=True)  # :57
    else:
        x1 = F.conv_transpose2d(x1, sd[p + "up.weight"], sd[p + "up.bias"], stride=2)  # :63-64
    x1 = _pad_to(x1, x2)
    x = x2 * x1 if use_attention else torch.cat([x2, x1], dim=1)                     # :84-87
    return double_conv(sd, p + "conv.", x, train)


def out_conv(sd: SD, p: str, x: Tensor) -> Tensor:
    """OutConv.forward, model_parts.py:93-99: 1x1 conv."""
    return F.conv2d(x, sd[p + "conv.weight"], sd[p + "conv.bias"])


# --------------------------------------------------------------------------------------------
# networks (models.py)
# --------------------------------------------------------------------------------------------
def unet_forward(sd: SD, x: Tensor, train: bool = True, bilinear: bool = False,
                 use_attention: bool = False) -> Tensor:
    """UNet.forward, models.py:53-68."""
    x1 = double_conv(sd, "inc.", x, train)
    x2 = down(sd, "down1.", x1, train)
    x3 = down(sd, "down2.", x2, train)
    x4 = down(sd, "down3.", x3, train)
    x5 = down(sd, "down4.", x4, train)
    y = up(sd, "up1.", x5, x4, train, bilinear, use_attention)
    y = up(sd, "up2.", y, x3, train, bilinear, use_attention)
    y = up(sd, "up3.", y, x2, train, bilinear, use_attention)
    y = up(sd, "up4.", y, x1, train, bilinear, use_attention)
    return out_conv(sd, "outc.", y)


def cubenet_forward(sd: SD, x: Tensor, first_depth: int = 64, train: bool = True,
                    conv3d: bool = True, bilinear: bool = False, use_attention: bool = False) -> Tensor:
    """CubeNET.forward, models.py:202-247 (bilinear=False, use_attention=False -- the configured
    path, params_HyperPRI.py:210-211).  ``conv3d=False`` evaluates the first layer as the
    algebraically identical Conv2d over D input channels."""
    n, _, d, h, w = x.shape
    w0 = sd["first_conv.weight"]
    if conv3d:
        x1 = F.conv3d(x, w0, sd["first_conv.bias"], padding=(0, 1, 1))               # :169,215
        x1 = F.relu(_bn(sd, "inc.1", x1, train))                                     # :172-173
        x1 = x1.reshape(n, w0.shape[0], h, w)                                        # :216
    else:
        x1 = F.conv2d(x.reshape(n, d, h, w), w0.reshape(w0.shape[0], d, 3, 3),
                      sd["first_conv.bias"], padding=1)
        x1 = F.relu(_bn(sd, "inc.1", x1, train))
    x1 = F.conv2d(x1, sd["inc2.0.weight"], sd["inc2.0.bias"], padding=1)            # :176-180
    x1 = F.relu(_bn(sd, "inc2.1", x1, train))
    x2 = down(sd, "down1.", x1, train)
    x3 = down(sd, "down2.", x2, train)
    x4 = down(sd, "down3.", x3, train)
    x5 = down(sd, "down4.", x4, train)
    y = up(sd, "up1.", x5, x4, train, bilinear, use_attention)
    y = up(sd, "up2.", y, x3, train, bilinear, use_attention)
    y = up(sd, "up3.", y, x2, train, bilinear, use_attention)
    if first_depth == 64:
        y = up(sd, "up4.", y, x1, train, bilinear, use_attention)                    # :228
    else:                                                                            # :229-240
        if bilinear:
            y = F.interpolate(y, scale_factor=2, mode="bilinear", align_corners=True)
        else:
            y = F.conv_transpose2d(y, sd["upsample4.weight"], sd["upsample4.bias"], stride=2)
        y = _pad_to(y, x1)
        y = torch.cat([x1, y], dim=1)
        y = double_conv(sd, "upconv4.", y, train)
    return out_conv(sd, "outc.", y)


def _basic(sd: SD, p: str, x: Tensor, train: bool) -> Tensor:
    """SpectralUNET._basic_module, models.py:105-115: Linear -> BatchNorm1d -> ReLU; with ``bnorm=False`` (:106-110) the module is
    Linear -> ReLU and its state dict has no ``<name>.1.*`` entries."""
    x = F.linear(x, sd[p + ".0.weight"], sd[p + ".0.bias"])
    if (p + ".1.weight") not in sd:
        return F.relu(x)
    return F.relu(_bn(sd, p + ".1", x, train))


def spectral_forward(sd: SD, x: Tensor, train: bool = True) -> Tensor:
    """SpectralUNET.forward, models.py:117-145.  One image at a time (``for idx, in_x``, :132), so
    BatchNorm1d statistics are per image and running stats advance N times per call."""
    n, d, r, c = x.shape
    n_classes = sd["outc.weight"].shape[0]
    rast = x.reshape(n, d, r * c).permute(0, 2, 1)                                   # :130
    outs = []
    for i in range(n):
        x0 = _basic(sd, "tail", rast[i], train)
        x1 = _basic(sd, "down1", x0, train)
        x2 = _basic(sd, "down2", x1, train)
        x3 = _basic(sd, "down3", x2, train)
        x4 = _basic(sd, "down4", x3, train)
        t = _basic(sd, "up1", x4, train)
        t = _basic(sd, "up2", torch.cat((x3, t), -1), train)                         # :139
        t = _basic(sd, "up3", torch.cat((x2, t), -1), train)
        t = _basic(sd, "up4", torch.cat((x1, t), -1), train)
        t = F.linear(torch.cat((x0, t), -1), sd["outc.weight"], sd["outc.bias"])    # :143
        outs.append(t.reshape(n_classes, r, c))                                      # :144
    return torch.stack(outs, 0)


# --------------------------------------------------------------------------------------------
# state-dict construction (key order == the reference's registration order)
# --------------------------------------------------------------------------------------------
def _conv_keys(sd, p, cout, cin, k, dims=2):
    sd[p + ".weight"] = (cout, cin) + (k,) * dims
    sd[p + ".bias"] = (cout,)


def _bn_keys(sd, p, c):
    sd[p + ".weight"] = (c,)
    sd[p + ".bias"] = (c,)
    sd[p + ".running_mean"] = (c,)
    sd[p + ".running_var"] = (c,)
    sd[p + ".num_batches_tracked"] = ()


def _dc_keys(sd, p, cin, cout, mid=None):
    mid = mid or cout
    _conv_keys(sd, p + "double_conv.0", mid, cin, 3)
    _bn_keys(sd, p + "double_conv.1", mid)
    _conv_keys(sd, p + "double_conv.3", cout, mid, 3)
    _bn_keys(sd, p + "double_conv.4", cout)


def _up_keys(sd, p, cin, cout, bilinear=False, attention=False):
    """Up.__init__, model_parts.py:51-68."""
    if bilinear:                                         # nn.Upsample has no parameters
        _dc_keys(sd, p + "conv.", cin // 2 if attention else cin, cout // 2, cin // 2)
    else:
        sd[p + "up.weight"] = (cin, cin // 2, 2, 2)      # ConvTranspose2d layout (Cin,Cout,kh,kw)
        sd[p + "up.bias"] = (cin // 2,)
        _dc_keys(sd, p + "conv.", cin // 2 if attention else cin, cout)


def unet_shapes(n_channels: int, n_classes: int, bilinear: bool = False,
                use_attention: bool = False) -> "OrderedDict[str, tuple]":
    """state_dict keys/shapes of UNet(n_channels, n_classes, bilinear, use_attention), models.py:24-51."""
    f = 2 if bilinear else 1
    sd = OrderedDict()
    _dc_keys(sd, "inc.", n_channels, 64)
    for i, (a, b) in enumerate([(64, 128), (128, 256), (256, 512), (512, 1024 // f)], 1):
        _dc_keys(sd, f"down{i}.maxpool_conv.1.", a, b)
    for i, (a, b) in enumerate([(1024, 512), (512, 256), (256, 128), (128, 64 * f)], 1):
        _up_keys(sd, f"up{i}.", a, b, bilinear, use_attention)
    _conv_keys(sd, "outc.conv", n_classes, 64, 1)
    return sd


def cubenet_shapes(depth: int, n_classes: int, first_depth: int = 64, bilinear: bool = False,
                   use_attention: bool = False) -> "OrderedDict[str, tuple]":
    """state_dict keys/shapes of CubeNET(depth, n_classes, first_depth, bilinear, use_attention),
    models.py:149-200.  ``first_conv.*`` and ``inc.0.*`` alias one tensor (models.py:169-171)."""
    f = 2 if bilinear else 1
    sd = OrderedDict()
    _conv_keys(sd, "first_conv", first_depth, 1, 3)
    sd["first_conv.weight"] = (first_depth, 1, depth, 3, 3)
    sd["inc.0.weight"] = sd["first_conv.weight"]
    sd["inc.0.bias"] = sd["first_conv.bias"]
    _bn_keys(sd, "inc.1", first_depth)
    _conv_keys(sd, "inc2.0", first_depth, first_depth, 3)
    _bn_keys(sd, "inc2.1", first_depth)
    for i, (a, b) in enumerate([(first_depth, 128), (128, 256), (256, 512), (512, 1024 // f)], 1):
        _dc_keys(sd, f"down{i}.maxpool_conv.1.", a, b)
    for i, (a, b) in enumerate([(1024, 512), (512, 256), (256, 128)], 1):
        _up_keys(sd, f"up{i}.", a, b, bilinear, use_attention)
    if first_depth == 64:
        _up_keys(sd, "up4.", 128, 64 * f, bilinear, use_attention)
    elif bilinear:
        _dc_keys(sd, "upconv4.", 128 + first_depth, 64, 64)
    else:
        sd["upsample4.weight"] = (128, 64, 2, 2)
        sd["upsample4.bias"] = (64,)
        _dc_keys(sd, "upconv4.", 64 + first_depth, 64)
    _conv_keys(sd, "outc.conv", n_classes, 64, 1)
    return sd


def spectral_shapes(depth: int, n_classes: int, f: int, bnorm: bool = True) -> "OrderedDict[str, tuple]":
    """state_dict keys/shapes of SpectralUNET(depth, n_classes, bn_feats=f, bnorm=bnorm), models.py:72-103."""
    sd = OrderedDict()
    for name, cin in [("tail", depth), ("down1", f), ("down2", f), ("down3", f), ("down4", f),
                      ("up1", f), ("up2", 2 * f), ("up3", 2 * f), ("up4", 2 * f)]:
        sd[name + ".0.weight"] = (f, cin)
        sd[name + ".0.bias"] = (f,)
        if bnorm:
            _bn_keys(sd, name + ".1", f)
    sd["outc.weight"] = (n_classes, 2 * f)
    sd["outc.bias"] = (n_classes,)
    return sd


def _u(seed: int, count: int) -> np.ndarray:
    """Same counter-based generator as hyperpri_amd/synth.py (restated so the oracle does not
    import the product): u = (splitmix64_mix(seed*GOLDEN + idx) >> 40) / 2^24."""
    g, m1, m2 = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xBF58476D1CE4E5B9), np.uint64(0x94D049BB133111EB)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * g + np.arange(count, dtype=np.uint64)
        z ^= z >> np.uint64(30); z *= m1
        z ^= z >> np.uint64(27); z *= m2
        z ^= z >> np.uint64(31)
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)


def synth_state_dict(shapes: "OrderedDict[str, tuple]", seed0: int = 1000,
                     bn_random: bool = False) -> SD:
    """Generator-defined weights: the k-th *parameter* in registration order
    is (2u(seed0+k, .)-1)/sqrt(fan_in); a bias uses its layer's weight fan_in (PyTorch default
    bound); BN gamma=1, beta=0, running_mean=0, running_var=1, num_batches_tracked=0.
    ``bn_random`` (block fixtures only) draws gamma = 0.5+u, beta = u-0.5 instead."""
    sd: SD = OrderedDict()
    k = 0
    last_fan_in = 1
    for name, shp in shapes.items():
        if name.startswith("inc.0."):            # alias of first_conv.* (one Parameter)
            sd[name] = sd["first_conv." + name.split(".")[-1]]
            continue
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            sd[name] = torch.zeros((), dtype=torch.int64)
        elif leaf == "running_mean":
            sd[name] = torch.zeros(shp)
        elif leaf == "running_var":
            sd[name] = torch.ones(shp)
        elif (name.rsplit(".", 1)[0] + ".running_mean") in shapes:
            # BatchNorm affine parameters: gamma = 1, beta = 0
            if bn_random:
                v = _u(seed0 + k, int(np.prod(shp))) + np.float32(0.5 if leaf == "weight" else -0.5)
                sd[name] = torch.from_numpy(v.reshape(shp).copy())
            else:
                sd[name] = torch.ones(shp) if leaf == "weight" else torch.zeros(shp)
            k += 1
        else:
            if len(shp) >= 2:
                rf = 1
                for s in shp[2:]:
                    rf *= s
                last_fan_in = shp[1] * rf
            cnt = int(np.prod(shp))
            v = (2.0 * _u(seed0 + k, cnt) - 1.0) * np.float32(1.0 / math.sqrt(last_fan_in))
            sd[name] = torch.from_numpy(v.astype(np.float32).reshape(shp).copy())
            k += 1
    return sd


def is_param(name: str) -> bool:
    leaf = name.rsplit(".", 1)[-1]
    return leaf in ("weight", "bias") and not name.startswith("inc.0.")


# --------------------------------------------------------------------------------------------
# loss / metrics restated (caller side: PLTrainer.py:79-98, 62-68)
# --------------------------------------------------------------------------------------------
def bce_with_logits(logits: Tensor, mask: Tensor) -> Tensor:
    """nn.BCEWithLogitsLoss() (mean) -- params_HyperPRI.py:60,223."""
    return F.binary_cross_entropy_with_logits(logits, mask)


def seg_metrics(logits: Tensor, mask: Tensor, thr: float = 0.5) -> Tuple[float, float, float]:
    """(pixel accuracy, positive-class Dice, positive IoU) as PLTrainer.py:88-91 obtains them from
    torchmetrics: seg = sigmoid(logits) > thr; Dice = 2TP/(2TP+FP+FN) (Dice(num_classes=2,
    ignore_index=0, zero_division=1e-12) :64-67); IoU = TP/(TP+FP+FN) (JaccardIndex binary :63)."""
    seg = torch.sigmoid(logits.detach().double()) > thr
    m = mask > 0.5
    tp = float((seg & m).sum()); fp = float((seg & ~m).sum()); fn = float((~seg & m).sum())
    tn = float((~seg & ~m).sum())
    acc = (tp + tn) / max(tp + tn + fp + fn, 1.0)
    dice = (2 * tp) / (2 * tp + fp + fn) if (2 * tp + fp + fn) > 0 else 1e-12
    iou = tp / (tp + fp + fn) if (tp + fp + fn) > 0 else 0.0
    return acc, dice, iou


def train_step(forward, sd: SD, x: Tensor, mask: Tensor, **kw):
    """One ``training_step`` (PLTrainer.py:79-98) minus the optimizer: logits, BCE loss and the
    gradient of every parameter.  Returns (logits, loss, {name: grad})."""
    leaves = OrderedDict()
    work: SD = OrderedDict()
    for k, v in sd.items():
        if is_param(k):
            t = v.detach().clone().requires_grad_(True)
            leaves[k] = t
            work[k] = t
        elif k.startswith("inc.0."):
            work[k] = work["first_conv." + k.split(".")[-1]]
        else:
            work[k] = v  # buffers are updated in place, as the modules do
    logits = forward(work, x, **kw)
    loss = bce_with_logits(logits, mask)
    loss.backward()
    return logits.detach(), float(loss.detach()), OrderedDict((k, t.grad) for k, t in leaves.items())


# --------------------------------------------------------------------------------------------
# caller-side tail of a step: counts, binned PR curve, optimizer
# --------------------------------------------------------------------------------------------
def seg_counts(pred: Tensor, mask: Tensor, thr: float = 0.5, is_logits: bool = True) -> Tuple[int, int, int, int]:
    """(TP, FP, FN, TN) of ``torch.sigmoid(pred) > thr`` in fp32, as PLTrainer.py:88 evaluates it, against
    ``mask.to(torch.int32)`` (PLTrainer.py:80)."""
    p = torch.sigmoid(pred.detach().float()) if is_logits else pred.detach().float()
    seg = (p > thr).flatten()
    m = (mask.to(torch.int32) != 0).flatten()
    return (int((seg & m).sum()), int((seg & ~m).sum()), int((~seg & m).sum()), int((~seg & ~m).sum()))


def pr_curve_binned(probs: Tensor, target: Tensor, thresholds: int = 500):
    """torchmetrics 1.2.0 ``PrecisionRecallCurve('binary', thresholds=T)`` (PLTrainer.py:542-543; torchmetrics is a
    pinned dependency, environment.yml:31, NOT installed here -> restated from its published algorithm, parity
    unpinned): thresholds = linspace(0, 1, T); per threshold the confusion matrix of (probs >= t) vs target;
    precision = tp/(tp+fp), recall = tp/(tp+fn) with 0 where the denominator is 0; a final (1, 0) point appended.
    Returns (precision[T+1], recall[T+1], thresholds[T], tp[T], fp[T], fn[T])."""
    thr = torch.linspace(0, 1, thresholds, dtype=torch.float32)
    p = probs.detach().float().flatten()
    t = (target.flatten().to(torch.int32) != 0)
    tp = torch.empty(thresholds, dtype=torch.int64)
    fp = torch.empty(thresholds, dtype=torch.int64)
    for k in range(thresholds):
        ge = p >= thr[k]
        tp[k] = int((ge & t).sum())
        fp[k] = int((ge & ~t).sum())
    fn = int(t.sum()) - tp

    def safe_div(a, b):
        b = torch.where(b == 0, torch.ones_like(b), b)
        return a / b

    tpf, fpf, fnf = tp.float(), fp.float(), fn.float()
    precision = torch.cat([safe_div(tpf, tpf + fpf), torch.ones(1)])
    recall = torch.cat([safe_div(tpf, tpf + fnf), torch.zeros(1)])
    return precision, recall, thr, tp, fp, fn


def best_dice_threshold(precision: Tensor, recall: Tensor, thresholds: Tensor):
    """PLTrainer.py:546-556: crop 1 % at both ends, Dice = 2PR/(P+R), arg-max, threshold rounded to 2 decimals."""
    crop = int(len(precision) // 100)
    p, r, t = precision[crop:-crop], recall[crop:-crop], thresholds[crop:-crop]
    dice = 2 * p * r / (p + r)
    i = int(torch.argmax(dice))
    return float(torch.round(t[i].to(torch.float), decimals=2)), float(p[i]), float(r[i])


def optimizer_steps(kind: str, params, grads_per_step, **hyper):
    """Run ``torch.optim.Adam`` / ``torch.optim.SGD`` (the optimizers PLTrainer.py:171-181 constructs) on CPU copies of
    ``params`` with the given per-step gradient lists; returns the updated tensors."""
    ps = [torch.nn.Parameter(p.detach().clone().float()) for p in params]
    opt = (torch.optim.Adam if kind.lower() == "adam" else torch.optim.SGD)(ps, **hyper)
    for grads in grads_per_step:
        for p, g in zip(ps, grads):
            p.grad = None if g is None else g.detach().clone().float()
        opt.step()
    return [p.detach() for p in ps]


def average_precision(probs: Tensor, target: Tensor) -> float:
    """``AveragePrecision(task='binary')`` (PLTrainer.py:558-559): the step-wise sum sum_n (R_n - R_{n-1}) P_n that
    torchmetrics shares with scikit-learn -- computed here by scikit-learn itself (installed; torchmetrics is not)."""
    from sklearn.metrics import average_precision_score
    return float(average_precision_score((target.flatten().to(torch.int32) != 0).numpy(), probs.detach().float().flatten().numpy()))
