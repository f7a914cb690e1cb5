"""Drop-in replacements for the reference's U-Net building blocks (src/Experiments/model_parts.py).

Same class names, constructor signatures, sub-module names and registration order as the reference
(``DoubleConv`` :14-31, ``Down`` :34-45, ``Up`` :48-90, ``OutConv`` :93-99), so ``state_dict`` keys,
default initialisation under ``torch.manual_seed`` and checkpoint loading are unchanged.  The
``torch.nn`` layers inside are only parameter containers: ``forward`` runs hand-written HIP kernels
(fp32 MFMA implicit-GEMM convs, fused BN+ReLU, max-pool, transposed-conv scatter into the concat
buffer) through ``engine``/``autograd`` and never calls a torch compute op.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F  # re-exported: the reference star-imports this module (models.py:14)

from . import engine as E
from .autograd import run

__all__ = ["DoubleConv", "Down", "Up", "OutConv", "torch", "nn", "F", "set_precision"]     # (skip_room, has_hooks: internal)


def _drain(gen):
    """Run a staged tape program (a generator that yields at its stage boundaries: autograd.run_staged) to its end."""
    try:
        while True:
            next(gen)
    except StopIteration as stop:
        return stop.value


def _double_conv_gen(tape, x, seq, train, need_dx=True, precision=None, room=0, out_planes=False, head_next=False):
    """(conv3x3 -> BN -> ReLU) x 2 on an Act, as a staged program (one stage per convolution); ``seq`` is the 6-entry
    nn.Sequential container.  ``room``: channels to keep free behind the result (it is a skip tensor: the decoder's concat is
    then in place).  ``out_planes``: the result feeds a transposed convolution that reads bf16 planes (bf16 mode): its
    BatchNorm-apply pass writes them.  ``head_next``: the result is read by the 1x1 output layer only (engine.conv_bn_relu)."""
    h = E.conv_bn_relu(tape, x, seq[0].weight, seq[0].bias, E.BNRef(seq[1]), train, 3, need_dx=need_dx, precision=precision,
                       next_cout=seq[3].weight.shape[0])       # h is read by the second convolution only
    yield
    return E.conv_bn_relu(tape, h, seq[3].weight, seq[3].bias, E.BNRef(seq[4]), train, 3, precision=precision, room=room,
                          out_planes=out_planes, head_next=head_next)


def _double_conv_ops(*args, **kw):
    return _drain(_double_conv_gen(*args, **kw))


def skip_room(up):
    """Channels the decoder stage ``up`` (an ``Up`` module, or a bare ConvTranspose2d) will append behind its skip tensor:
    known only for the transposed-convolution concat path; 0 = let ``up_concat`` copy."""
    if isinstance(up, nn.ConvTranspose2d):
        return up.out_channels
    if isinstance(up, Up) and not up.bilinear and not up.use_attention:
        return up.up.out_channels
    return 0


def has_hooks(module):
    """True when any sub-module carries forward hooks: the networks then call their children one by one (each its own
    autograd node, hooks fire as in the reference) instead of running as one fused tape."""
    for m in module.modules():
        if m is not module and (m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks):
            return True
    return False


def set_precision(module, precision):
    """Select the contraction arithmetic for ``module`` and all its children: "fp32" (exact, the default: the reference's semantics),
    "bf16" / "f16" (16-bit MFMA operands and 16-bit stored activations: bf16, or IEEE half -- 11 mantissa bits instead of 8, the
    kernels of libhyperpri_hip_f16.so, activation gradients under a power-of-two loss scale chosen from the number of logits of a
    mean-reduced loss), "bf16x3" / "bf16x6" (fp32 operands split into 2 / 3 bf16 planes).  Not part of the reference API.
    "f16" is a mode of whole networks (UNet / CubeNET / SpectralUNET): the loss scale enters at their output layer."""
    if precision not in E.PRECISIONS + ("f16",):
        raise ValueError(f"precision must be one of {E.PRECISIONS + ('f16',)}")
    for m in module.modules():
        m.hpri_precision = "bf16" if precision == "f16" else precision      # (the engine's plane paths; the 16-bit TYPE is the library's)
        m.hpri_h16 = "f16" if precision == "f16" else None
    return module


class DoubleConv(nn.Module):
    """(convolution => [BN] => ReLU) * 2 -- reference model_parts.py:14-31."""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        mid = mid_channels if mid_channels else out_channels
        layers = [nn.Conv2d(in_channels, mid, kernel_size=3, padding=1), nn.BatchNorm2d(mid), nn.ReLU(inplace=True),
                  nn.Conv2d(mid, out_channels, kernel_size=3, padding=1), nn.BatchNorm2d(out_channels),
                  nn.ReLU(inplace=True)]
        self.double_conv = nn.Sequential(*layers)

    def _gen(self, tape, x, need_dx=True, room=0, out_planes=False, head_next=False):
        return _double_conv_gen(tape, x, self.double_conv, self.training, need_dx, getattr(self, "hpri_precision", None), room,
                                out_planes, head_next)

    def _ops(self, *args, **kw):
        return _drain(self._gen(*args, **kw))

    def _stages(self):
        """Modules per stage of ``_gen`` (whose parameters' gradients are final once backward has passed the stage)."""
        seq = self.double_conv
        return [[seq[0], seq[1]], [seq[3], seq[4]]]

    def forward(self, x):
        return run(lambda tape, a, need: self._ops(tape, a[0], need[0]), [x], list(self.parameters()), name="double_conv",
                   lib_kind=getattr(self, "hpri_h16", None))


class Down(nn.Module):
    """MaxPool2d(2) then DoubleConv -- reference model_parts.py:34-45."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels))

    def _gen(self, tape, x, room=0, out_planes=False):
        return self.maxpool_conv[1]._gen(tape, E.maxpool2(tape, x), room=room, out_planes=out_planes)

    def _ops(self, *args, **kw):
        return _drain(self._gen(*args, **kw))

    def _stages(self):
        return self.maxpool_conv[1]._stages()

    def forward(self, x):
        return run(lambda tape, a, need: self._ops(tape, a[0]), [x], list(self.parameters()), name="down", lib_kind=getattr(self, "hpri_h16", None))


class Up(nn.Module):
    """up -> zero-pad to the skip -> cat([skip, up]) (or skip*up when ``use_attention``) -> DoubleConv
    (reference model_parts.py:48-90).  ``up`` is ConvTranspose2d(k2,s2) for ``bilinear=False`` -- the path every
    HyperPRI experiment configures (params_HyperPRI.py:53-55,210-211) -- or bilinear x2 upsampling."""

    def __init__(self, in_channels, out_channels, bilinear=True, use_attention=False):
        super().__init__()
        self.use_attention = use_attention
        self.bilinear = bilinear
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
            if use_attention:
                self.conv = DoubleConv(in_channels // 2, out_channels // 2, in_channels // 2)
            else:
                self.conv = DoubleConv(in_channels, out_channels // 2, in_channels // 2)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(in_channels // 2 if use_attention else in_channels, out_channels)

    def _gen(self, tape, x1, x2, need_dx1=True, out_planes=False, head_next=False):
        w = None if self.bilinear else self.up.weight
        b = None if self.bilinear else self.up.bias
        join = E.up_attention if self.use_attention else E.up_concat
        return self.conv._gen(tape, join(tape, x1, x2, w, b, need_dx1=need_dx1,
                                         precision=getattr(self, "hpri_precision", None)), out_planes=out_planes, head_next=head_next)

    def _ops(self, *args, **kw):
        return _drain(self._gen(*args, **kw))

    def _stages(self):
        first, second = self.conv._stages()
        return [[self.up] + first, second]            # (nn.Upsample has no parameters)

    def forward(self, x1, x2):
        return run(lambda tape, a, need: self._ops(tape, a[0], a[1], need[0]), [x1, x2], list(self.parameters()), name="up",
                   lib_kind=getattr(self, "hpri_h16", None))


class OutConv(nn.Module):
    """1x1 conv to the class logits -- reference model_parts.py:93-99."""

    def __init__(self, in_channels, out_channels):
        super(OutConv, self).__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def _ops(self, tape, x, need_dx=True):
        return E.out_conv(tape, x, self.conv.weight, self.conv.bias, need_dx)

    def forward(self, x):
        return run(lambda tape, a, need: self._ops(tape, a[0], need[0]), [x], list(self.parameters()), name="out_conv",
                   lib_kind=getattr(self, "hpri_h16", None))
