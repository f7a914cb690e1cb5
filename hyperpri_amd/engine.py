"""Host-side engine: NHWC activation views, a reverse-mode tape, and one Python wrapper per HIP op.

Everything numeric happens in libhyperpri_hip.so (see include/hyperpri_hip.h); PyTorch is used for
device memory (caching allocator), the current HIP stream and autograd plumbing only.  The op
wrappers mirror what the reference's layers do (file:line cited per op) and record a backward
closure on the tape; ``model_parts.py`` / ``models.py`` compose them into the reference's modules.
"""
from __future__ import annotations

import contextlib
import ctypes
import threading
import os
import types
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _lib

A_DIRECT, A_S2D = 0, 1
E_DIRECT, E_D2S = 0, 1

# Arithmetic of the conv / linear contractions:
#   "fp32" (default) -- exact fp32 on v_mfma_f32_32x32x2_f32: the mode every 1e-3-logit parity claim refers to
#   "bf16"           -- conv / linear operands (forward, data gradient, weight gradient) rounded to bf16,
#                       v_mfma_f32_32x32x16_bf16 with fp32 accumulate (BASELINE.json config C5); activations in HBM,
#                       BN, pooling and the 1x1 output conv stay fp32.
#   "bf16x3"         -- every contraction operand carried as bf16 hi + bf16 lo (16 mantissa bits), three bf16 MFMAs per
#                       product (hi*hi + hi*lo + lo*hi), fp32 accumulate: ~4e-5 on the logits, i.e. INSIDE the 1e-3
#                       contract, on the 16x faster pipe.
#   "bf16x6"         -- three planes hi + mid + lo = 24 mantissa bits, i.e. every fp32 operand EXACTLY; six bf16 MFMAs per
#                       product (the three cross terms below 2^-24 are dropped), fp32 accumulate: fp32-class accuracy
#                       (what cuBLAS calls fp32 emulation) on the bf16 pipe.
#                       Dice/IoU-level parity only (bf16 operands move logits by ~2e-2).
PRECISIONS = ("fp32", "bf16", "bf16x3", "bf16x6")
LOWP = ("bf16", "bf16x3", "bf16x6")
_SPLIT = {"bf16": 0, "bf16x3": 1, "bf16x6": 2}
DEFAULT_PRECISION = os.environ.get("HPRI_PRECISION", "fp32")


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


# Item queues of the persistent MFMA kernels (include/hyperpri_hip.h: hpri_set_item_queue): one zeroed counter buffer per stream the
# engine launches on, registered the first time the stream is seen.  With a queue a workgroup that becomes resident late (an RCCL
# kernel holds part of its CU during a DDP backward) finds the items gone instead of walking a fixed list alone; results do not
# change.  engine.ITEM_QUEUE = False before the first launch: fixed lists (the round-4 behaviour; tools/cu_share_probe.py measures both).
ITEM_QUEUE = True
_item_queues: Dict[tuple, torch.Tensor] = {}


def _stream() -> ctypes.c_void_p:
    st = torch.cuda.current_stream()
    h = st.cuda_stream
    key = (st.device_index, h) if _lib.kind() == "bf16" else (st.device_index, h, _lib.kind())      # (each library keeps its own registry)
    if ITEM_QUEUE and key not in _item_queues:
        lib = _lib.current()
        q = torch.zeros(lib.hpri_item_queue_bytes() // 4, dtype=torch.int32, device=torch.device("cuda", st.device_index))
        _item_queues[key] = q           # (kept for the life of the process: the library holds the raw pointer)
        _lib.call("hpri_set_item_queue", _p(q), q.numel() * 4, ctypes.c_void_p(h))
    return ctypes.c_void_p(h)


def set_plan_option(name: str, value: int) -> None:
    """A launch-plan option of the HIP library (include/hyperpri_hip.h: hpri_set_option), e.g. "wgrad_cu_reserve"; set in both builds
    of the library (the f16 mode's launches go to the half-precision one, which keeps its own options)."""
    for kind in (None, "f16"):
        if kind == "f16" and not os.path.exists(_lib.LIB_F16_PATH):
            continue
        with _lib.using(kind):
            _lib.call("hpri_set_option", name.encode(), int(value))


def scale_tensors_(tensors: List[torch.Tensor], scale: float) -> None:
    """t *= scale for every (contiguous fp32) tensor of the list, one launch per 48 tensors (hpri_scale_tensors)."""
    ts = [t for t in tensors if t is not None and t.numel() > 0]
    if not ts or scale == 1.0:
        return
    for t in ts:
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("hyperpri_amd: internal error: scale_tensors_ wants contiguous fp32 tensors")
    ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    n = (ctypes.c_longlong * len(ts))(*[t.numel() for t in ts])
    _lib.call("hpri_scale_tensors", ptrs, n, len(ts), float(scale), _stream())


def _p(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _require_cuda(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"hyperpri_amd: {what} is on {t.device}; the hot path exists only as HIP kernels for "
            "MI355X (no CPU fallback). Move the module and its inputs to a ROCm device.")
    if t.dtype != torch.float32:
        raise RuntimeError(f"hyperpri_amd: {what} must be float32, got {t.dtype}")


class Act:
    """fp32 NHWC activation view: element (n,h,w,c) at buf[((n*H+h)*W+w)*cs + coff + c].

    Invariant: channels [C, cw) (cw = C rounded up to 8) exist inside the stride and hold zeros, so
    consumers may run their K loop over ``cw`` channels."""
    __slots__ = ("buf", "N", "H", "W", "C", "cs", "coff", "pl", "parent", "f32_valid", "want_pl", "pl_part", "colsum_req", "b16", "bn_src", "cat_pl", "up_slice", "yr16", "skip_g16", "raw")

    def __init__(self, buf: torch.Tensor, N: int, H: int, W: int, C: int, cs: int, coff: int = 0):
        self.buf, self.N, self.H, self.W, self.C, self.cs, self.coff = buf, N, H, W, C, cs, coff
        self.pl: Optional["Planes"] = None      # bf16 plane copy of this activation (bf16 precision modes), see planes_of
        self.parent: Optional["Act"] = None     # wider buffer this Act is the leading channel slice of (new_with_room)
        self.f32_valid = True                   # False: only the bf16 planes were written (plane mode, inner tensor of a DoubleConv)
        self.pl_part = None                     # (Planes, channels filled so far): a concat buffer whose skip half is already in planes
        self.want_pl = 0                        # plane mode marker: planes a 3x3 consumer of this tensor (or of its pooled map) would read
        self.colsum_req = None                  # (c0, C): somebody wants the column sums of channels [c0, c0+C) of this tensor's GRADIENT
        self.b16 = False                        # the buffer holds bf16 elements (a pre-BN tensor of the bf16 mode; read by the *_x16 BN passes only)
        self.up_slice = None                    # (first channel, channels): the upsampled half of a decoder concat whose gradient its ConvTranspose2d wants as bf16 rows
        self.cat_pl = None                      # (plane buffer, cs, offset of the second half, its channels): this tensor's planes are the first half of a padded concat
        self.bn_src = None                      # (pre-BN Act, statistics, relu): this tensor is BN(+ReLU) of that one and has ONE consumer
        self.yr16 = False                       # the pre-BN tensor behind this one is stored as bf16 (its BatchNorm backward can read a bf16 gradient)
        self.skip_g16 = False                   # a decoder concat whose skip half's gradient is stored as bf16 rows (SKIP_GRAD_BF16)
        self.raw = None                         # (caller's NCHW tensor, planes for the layout pass): not laid out yet (Act.raw_nchw)

    @property
    def cw(self) -> int:
        return _rup(self.C, 8)

    @property
    def P(self) -> int:
        return self.N * self.H * self.W

    @property
    def ptr(self) -> ctypes.c_void_p:
        return ctypes.c_void_p(self.buf.data_ptr())

    @staticmethod
    def new(N: int, H: int, W: int, C: int, device) -> "Act":
        cs = _rup(C, 8)
        return Act(torch.empty(N * H * W * cs, dtype=torch.float32, device=device), N, H, W, C, cs, 0)

    @staticmethod
    def new_with_room(N: int, H: int, W: int, C: int, room: int, device) -> "Act":
        """An Act that is channels [0, C) of a fresh [N,H,W,C+room] buffer: a skip tensor written where the decoder's
        concat will need it (model_parts.py:87), so that ``up_concat`` has nothing to copy.  C % 8 == 0 keeps the
        zero-pad invariant trivially true for the slice."""
        if room <= 0 or C % 8:
            return Act.new(N, H, W, C, device)
        parent = Act.new(N, H, W, C + room, device)
        a = parent.slice(0, C)
        a.parent = parent
        return a

    def slice(self, c0: int, C: int) -> "Act":
        if c0 % 4:
            raise RuntimeError("hyperpri_amd: channel slices must start at a multiple of 4")
        v = Act(self.buf, self.N, self.H, self.W, C, self.cs, self.coff + c0)
        if self.b16:                       # a channel slice of bf16 rows (a gradient stored as bf16) is bf16 rows
            v.b16, v.f32_valid = True, False
        return v

    def to_tensor(self) -> torch.Tensor:
        """Logical (N,C,H,W) tensor aliasing this view (channels-last strides, no copy)."""
        return torch.as_strided(self.buf, (self.N, self.C, self.H, self.W),
                                (self.H * self.W * self.cs, 1, self.W * self.cs, self.cs), self.coff)

    @staticmethod
    def from_tensor(t: torch.Tensor, npl: int = 0) -> "Act":
        """(N,C,H,W) tensor -> Act.  Zero-copy when t is channels-last with a stride that already
        provides the zero pad; otherwise one nchw->nhwc HIP transpose (dataset.py:267-271 hands NCHW), which for
        ``npl`` > 0 also writes the bf16 planes the bf16-mode first convolution stages by DMA (``npl`` < 0: |npl| planes and
        no fp32 copy at all)."""
        _require_cuda(t, "input tensor")
        N, C, H, W = t.shape
        st = t.stride()
        cs = st[3] if W > 1 else (st[2] if H > 1 else C)
        # C % 8 != 0 is only zero-copy for tensors the ingest path built (ingest.py): it guarantees zero pad channels
        padded_ok = C % 8 == 0 or (getattr(t, "_hpri_zero_padded", False) and cs >= _rup(C, 8))
        if (padded_ok and st[1] == 1 and cs >= C and cs % 4 == 0 and (W == 1 or st[3] == cs)
                and (H == 1 or st[2] == W * cs) and (N == 1 or st[0] == H * W * cs)
                and t.data_ptr() % 16 == 0):
            a = Act(t, N, H, W, C, cs, 0)
            a.buf = t  # data_ptr() of the view is element (0,0,0,0)
            return a
        if not t.is_contiguous():
            t = t.contiguous()   # rare: arbitrary strides from the caller
        planes_only = npl < 0          # the caller's first layer and its weight gradient read bf16 planes and nothing else
        npl = abs(npl)
        a = Act.new(N, H, W, C, t.device)
        if npl > 0 and PLANE_PRODUCERS and (H * W) % 4 == 0 and t.data_ptr() % 16 == 0:
            pl = new_planes(a, npl)
            if planes_only and H * W * max(_rup(C, 32), 128) * 2 < 0x7FFFFF00:     # (as _planes_fit for a first layer <= 128 wide)
                a.f32_valid = False
            _lib.call("hpri_nchw_to_nhwc_pl", _p(t), a.ptr if a.f32_valid else ctypes.c_void_p(0), N, C, H * W, a.cs, 0, a.cw,
                      *_pl_args(pl), _stream())
        else:
            _lib.call("hpri_nchw_to_nhwc", _p(t), a.ptr, N, C, H * W, a.cs, 0, a.cw, _stream())
        return a

    @staticmethod
    def raw_nchw(t: torch.Tensor, npl: int) -> Optional["Act"]:
        """The caller's contiguous (N,C,H,W) fp32 tensor as an Act that has NOT been laid out: the predict path's first 3x3 layer
        reads it as it is (``_conv_ingest_eval``, csrc/conv_ingest.hip); any other reader calls ``materialize()`` = the layout pass
        ``from_tensor`` would have run.  None: not eligible (not contiguous NCHW, too few channels to matter, or > 2 GiB per image)."""
        if not (INGEST_FUSED and t.dim() == 4 and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 4 == 0):
            return None
        N, C, H, W = t.shape
        if C < 32 or H * W * C * 4 >= 0x7FFFFF00 or min(N, H, W) < 1:
            return None
        a = Act(t, N, H, W, C, _rup(C, 8), 0)
        a.raw, a.f32_valid = (t, npl), False
        return a

    def materialize(self) -> "Act":
        return Act.from_tensor(*self.raw) if self.raw is not None else self

    def to_nchw(self) -> torch.Tensor:
        out = torch.empty((self.N, self.C, self.H, self.W), dtype=torch.float32, device=self.buf.device)
        _lib.call("hpri_nhwc_to_nchw", self.ptr, _p(out), self.N, self.C, self.H * self.W, self.cs, self.coff, 0, _stream())
        return out


class Planes:
    """bf16 NHWC planes of an activation: plane p (0 = bf16(x), 1 = bf16(x - hi), ...) starts ``plane`` elements after
    the previous one, ``cs`` elements per pixel (a multiple of 32), channels [C, cs) are zero.  In the bf16 precision
    modes the 3x3 convolutions bring their operands into LDS by DMA straight from these planes (csrc/conv_bf16v2.hip)."""
    __slots__ = ("buf", "plane", "cs", "coff", "npl", "cw")

    def __init__(self, buf: torch.Tensor, plane: int, cs: int, coff: int, npl: int, cw: int = 0):
        self.buf, self.plane, self.cs, self.coff, self.npl = buf, plane, cs, coff, npl
        self.cw = cw or (cs - coff)          # channels of this view (a multiple of 32): narrower than cs - coff inside a padded concat


# fp32 mode: 3x3 convolutions (forward and data gradient) by Winograd F(2x2,3x3) on the fp32 MFMA (csrc/conv_wino.hip):
# 2.25x fewer multiplies in the same arithmetic type; 0 = direct implicit GEMM everywhere
WINOGRAD = os.environ.get("HPRI_WINOGRAD", "1") != "0"
WINO_WGRAD = WINOGRAD     # ... and the weight gradients (same switch)
WINO_MIN_BLOCKS = 256        # workgroups (16x16-pixel tiles x 64-channel blocks) below which the direct split-K kernel is used


def _wino_ok(x: "Act", ncols: int) -> bool:
    th = 8                                   # tile height of conv_wino4.hip (16 x 8 pixels)
    if x.H * x.W * x.cs * 4 >= (1 << 32) - 65536:      # 32-bit DMA offsets per image (conv_wino4 / wgrad): direct kernels beyond that
        return False
    return WINOGRAD and x.N * ((x.H + th - 1) // th) * ((x.W + 15) // 16) * (_rup(ncols, 64) // 64) >= WINO_MIN_BLOCKS


def _pack_wino(w: torch.Tensor, mode: int, K: int, ncols: int, d1: int) -> Tuple[torch.Tensor, int]:
    ncols_pad = _rup(ncols, 64)

    def build():
        global PACK_LAUNCHES
        up = torch.empty(_lib.load().hpri_wino_packed_floats(K, ncols_pad), dtype=torch.float32, device=w.device)
        _lib.call("hpri_wino4_pack", _p(w), _p(up), ctypes.c_void_p(0), mode, K, ncols, ncols_pad, d1, _stream())
        PACK_LAUNCHES += 1
        return up
    return _cached_pack(w, ("wino4", mode, K, ncols, d1), build), ncols_pad


def _conv_launch_wino(x: Act, up: torch.Tensor, bias: Optional[torch.Tensor], y: Act, stats: Optional[torch.Tensor],
                      cin: int, cout: int, cout_pad: int, y_cw: int, accumulate: int = 0) -> None:
    tag = "conv_winograd_f32<3,F(2x2)>"
    if SHAPE_TAGS:
        tag += f" N{x.N} {x.H}x{x.W} K{x.cw} N{cout}"
    wtiles = x.N * ((x.H + 1) // 2) * ((x.W + 1) // 2)
    with _timed(tag, 2.0 * x.N * x.H * x.W * cin * cout * 9, executed=2.0 * wtiles * 16 * cin * cout):
        _lib.call("hpri_conv_wino4", x.ptr, x.cs, x.coff, _p(up), _p(bias), y.ptr, y.cs, y.coff, _p(stats), x.N, x.H, x.W, x.cw,
                  cout, cout_pad, y_cw, accumulate, _stream())


# ---- switches ----------------------------------------------------------------------------------------------------------------------
# Twelve environment switches are documented: HPRI_PRECISION, HPRI_WINOGRAD, HPRI_SIDE_STREAM,
# HPRI_SIDE_STREAM_SINK, HPRI_PLANE_CONV, HPRI_PLANE_WGRAD, HPRI_PLANE_GEMM, HPRI_FUSIONS, HPRI_PACK_CACHE, HPRI_PACK_VERIFY,
# HPRI_STEPS_IN_FLIGHT, HPRI_DISPATCHER.  HPRI_FUSIONS=0 turns off, together, every traffic-saving fusion of rounds 2-3 (tensors
# kept as bf16 planes only, bf16 pre-BN tensors and inner gradients, plane concats, plane-fed transposed convolutions, BatchNorm /
# bias sums from a neighbour's epilogue): the results stay inside the same parity gates (most of the fusions are bit-neutral), the
# step reads and writes more.  The individual module attributes below exist so that tests can show each fusion's (non-)effect on
# the values one at a time; they have no environment variables of their own.
FUSIONS = os.environ.get("HPRI_FUSIONS", "1") != "0"
PLANE_CONV = os.environ.get("HPRI_PLANE_CONV", "1") != "0"   # bf16 mode: 3x3 convs on bf16 planes (0: round-1 kernel)
PLANE_WGRAD = os.environ.get("HPRI_PLANE_WGRAD", "1") != "0"  # ... and their weight gradients (0: round-1 kernel)
# the BatchNorm backward of a plane-mode layer writes its result as bf16 planes ONLY when both consumers read planes (one fp32
# tensor write less per layer and step).  (HPRI_FUSIONS.)
PLANES_ONLY_GRAD = FUSIONS
# ... and the inner tensor of a DoubleConv (conv -> BN -> ReLU -> [here] -> conv) likewise.  (HPRI_FUSIONS.)
PLANES_ONLY_ACT = FUSIONS
# no planes for tensors whose readers inside these networks are all fp32 (the output of a DoubleConv).  (HPRI_FUSIONS.)
PLANES_LAZY = FUSIONS
# a skip tensor's planes are written straight into the plane buffer of the decoder's concat.  (HPRI_FUSIONS.)
PLANES_CONCAT = FUSIONS
# ... and the transposed convolution writes its half of those planes itself (no fp32 form, no conversion).  (HPRI_FUSIONS.)
PLANES_CONVT = FUSIONS
PLANE_PRODUCERS = True       # producers (BN-apply, BN-backward, ...) write the planes themselves; False: generic pass only
PLANE_CONVERSIONS = 0        # generic fp32 -> planes passes launched (fused producers do not count)


# bf16 mode: the 1x1 layers (nn.Linear of SpectralUNET) forward and data gradient by the plane-fed GEMM kernel
# (gemm_bf16v3.hip); HPRI_PLANE_GEMM=0: the round-1 kernel that converts fp32 activations while staging.
PLANE_GEMM = os.environ.get("HPRI_PLANE_GEMM", "1") != "0"


# bf16 mode with the v3 plane convolution: the PRE-BatchNorm tensor of a conv -> BN -> ReLU stage (written by the convolution, read
# by the normalise pass and twice by the BatchNorm backward, by nobody else) is stored as bf16: 2 instead of 4 bytes per element on
# four tensor sweeps per layer and step.  The statistics still come from the fp32 accumulators.  Dice-level parity re-run:
# profiles/r03_bf16_dice_parity.json.  (HPRI_FUSIONS.)
YR_BF16 = FUSIONS


def _planes_fit(x: Act, channels: int) -> bool:
    """One image of bf16 planes of ``channels`` channels stays below the 2 GiB the plane kernels' DMA descriptors cover."""
    return x.H * x.W * _rup(channels, 32) * 2 < 0x7FFFFF00


def new_planes(x: Act, npl: int = 1) -> Planes:
    """Uninitialised plane storage for ``x`` (the producing kernel fills it, pad channels included)."""
    cs16 = _rup(x.C, 32)
    buf = torch.empty(npl * x.P * cs16, dtype=torch.bfloat16, device=x.buf.device)
    x.pl = Planes(buf, x.P * cs16, cs16, 0, npl)
    return x.pl


def _pl_args(pl: Optional[Planes]):
    """(planes, plane_stride, cs, coff, cw, npl) arguments of the *_pl entry points; all zero = fp32 only."""
    if pl is None:
        return ctypes.c_void_p(0), 0, 0, 0, 0, 0
    return _p(pl.buf), pl.plane, pl.cs, pl.coff, pl.cw, pl.npl


# Predict path of the 16-bit modes: the first 3x3 layer reads the caller's NCHW cube itself (csrc/conv_ingest.hip) instead of planes
# written by a layout pass.  ``INPUT_RAW_OK`` as a program's ``input_planes``: as -1 (one plane, no fp32 copy), and the program's first
# operation is a ``conv_bn_relu`` that may be handed the raw tensor when nothing is recorded (autograd._forward_impl).
INGEST_FUSED = True
INGEST_LAUNCHES = 0
INPUT_RAW_OK = -3


def input_planes_for(module, raw_ok: bool = False) -> int:
    """Planes the input layout pass should write for a network whose first layer is a 3x3 convolution (``raw_ok``: that layer is the
    program's first operation, a ``conv_bn_relu`` with one reader: see INPUT_RAW_OK)."""
    prec = getattr(module, "hpri_precision", None) or DEFAULT_PRECISION
    if not (PLANE_CONV and PLANE_PRODUCERS and prec == "bf16"):
        return 0
    if not (PLANES_ONLY_ACT and PLANE_WGRAD):
        return 1
    return INPUT_RAW_OK if (raw_ok and INGEST_FUSED) else -1      # -1: one plane and NO fp32 copy (Act.from_tensor)


def planes_of(x: Act, npl: int = 1) -> Planes:
    """The bf16 planes of ``x``; produced by the generic conversion pass unless the kernel that produced ``x`` has
    already written them."""
    global PLANE_CONVERSIONS
    if x.pl is not None and x.pl.npl >= npl:
        return x.pl
    cs16 = _rup(x.C, 32)
    buf = torch.empty(npl * x.P * cs16, dtype=torch.bfloat16, device=x.buf.device)
    _lib.call("hpri_to_planes", x.ptr, x.cs, x.coff, _p(buf), x.P * cs16, cs16, 0, x.P, x.C, cs16, npl, _stream())
    PLANE_CONVERSIONS += 1
    x.pl = Planes(buf, x.P * cs16, cs16, 0, npl)
    return x.pl


def _conv_launch_v2(x: Act, wp: torch.Tensor, bias: Optional[torch.Tensor], y: Act, stats: Optional[torch.Tensor],
                    cin: int, cout: int, cout_pad: int, y_cw: int, accumulate: int = 0) -> None:
    """3x3 convolution on bf16 planes (forward, or data gradient with the mode-1 pack)."""
    pl = planes_of(x, 1)
    cin_pad = _rup(cin, 32)
    ksplit = ctypes.c_int(); tiles = ctypes.c_int(); wsf = ctypes.c_size_t()
    _lib.call("hpri_conv_bf16v3_plan", x.N, x.H, x.W, cin_pad, cout_pad, ctypes.byref(ksplit), ctypes.byref(tiles), ctypes.byref(wsf))
    ws = _ws(wsf.value, x.buf.device) if wsf.value else None
    tag = "conv_planes_bf16<3,v3 256x64>"
    if SHAPE_TAGS:
        tag += f" N{x.N} {x.H}x{x.W} K{cin_pad} N{cout}"
    with _timed(tag, 2.0 * x.N * x.H * x.W * cin * cout * 9):
        _lib.call("hpri_conv_bf16v3", _p(pl.buf), pl.plane, pl.cs, pl.coff, _p(wp), _p(bias), y.ptr, y.cs, y.coff, _p(stats),
                  x.N, x.H, x.W, cin_pad, cout, cout_pad, y_cw, accumulate, 0, _p(ws), wsf.value, _stream())


def _gemm_launch(x: Act, wp: torch.Tensor, bias: Optional[torch.Tensor], y: Act, stats: Optional[torch.Tensor],
                 K: int, ncols: int, ncols_pad: int, y_cw: int, accumulate: int = 0) -> None:
    """1x1 convolution / Linear on bf16 planes (forward, or data gradient with the mode-1 pack): y may be an fp32 Act or a
    bf16 one (``b16``: the pre-BN tensor of the bf16 mode)."""
    pl = planes_of(x, 1)
    kpad = _rup(K, 32)
    null = ctypes.c_void_p(0)
    f32 = (y.ptr, y.cs, y.coff) if not y.b16 else (null, 0, 0)
    h16 = (y.ptr, y.cs, y.coff) if y.b16 else (null, 0, 0)
    tag = "gemm_planes_bf16<1,v3 256x128>"
    if SHAPE_TAGS:
        tag += f" N{x.N} {x.H}x{x.W} K{kpad} N{ncols}"
    with _timed(tag, 2.0 * x.P * K * ncols):
        _lib.call("hpri_gemm_bf16v3", _p(pl.buf), pl.cs, pl.coff, _p(wp), _p(bias), *f32, *h16, _p(stats), ncols_pad,
                  x.N, x.H * x.W, kpad, ncols, ncols_pad, y_cw, accumulate, _stream())


class Tape:
    """Reverse-mode tape over Acts.  ``nodes`` are closures run in reverse; ``grads`` maps id(Act) to
    the Act holding its gradient; parameter gradients are collected by id(parameter)."""

    def __init__(self, record: bool):
        self.record = record
        self.nodes: List[Callable[["Tape"], None]] = []
        self.grads: Dict[int, Act] = {}
        self.param_grads: Dict[int, torch.Tensor] = {}
        self.keep: List[object] = []
        self.side_keep: List[object] = []          # tensors the weight-gradient stream reads: alive until the caller has joined that stream (autograd.py)
        self.used_side = False
        self.sunk: Dict[int, torch.Tensor] = {}     # parameters whose gradient was written into the grad sink's storage
        self.colsum: Dict[int, tuple] = {}          # id(Act) -> (stats records, tiles, Cpad, c0) left by the data-gradient kernel that wrote its gradient
        self.gupl: Dict[int, tuple] = {}            # id(Act) -> (Planes of the gradient of a concat's upsampled half, fp32 form absent?)
        self.bnpart: Dict[int, tuple] = {}          # id(Act) -> (partial sums, blocks, Cpart) of its BatchNorm backward, left by the same kind of kernel
        self.uses: Dict[int, int] = {}              # id(parameter) -> ops recorded on this tape that will produce a gradient for it
        self._touched: List[int] = []               # parameters the running node asked a gradient slot for
        self.delivered: set = set()                 # segmented tape: parameters whose gradient has left with an earlier slice
        self.done_to: Optional[int] = None          # segmented tape: the backward has run down to this node index (a slice's node runs one stage ahead)
        self.gscale = 1.0                           # half-precision mode: the power of two the head multiplies into the gradient (out_conv)
        self._unscale: List[torch.Tensor] = []      # ... and the sunk gradients that still carry it (unscaled where their bucket is handed over)

    def note_params(self, *params: Optional[torch.Tensor]) -> None:
        """Called by an op while it records its backward node: it will contribute to these parameters' gradients.  A
        parameter used by several nodes (tied weights, a module called twice inside one tape) is handed to the gradient sink
        only after the LAST of them has run."""
        if self.record:
            for p in params:
                if p is not None and p.requires_grad:
                    self.uses[id(p)] = self.uses.get(id(p), 0) + 1

    def grad_slot(self, a: Act, b16_ok: bool = False) -> Tuple[Act, bool]:
        """(gradient view for ``a``, accumulate?) -- allocates a fresh buffer the first time.  ``b16_ok``: the caller can add into
        a gradient stored as bf16 rows (the pooling backward); everybody else writes fp32."""
        g = self.grads.get(id(a))
        if g is not None:
            if g.b16 and not b16_ok:
                raise RuntimeError("hyperpri_amd: internal error: an fp32 writer met a gradient stored as bf16 rows")
            return g, True
        g = Act.new(a.N, a.H, a.W, a.C, a.buf.device)
        self.grads[id(a)] = g
        return g, False

    def set_grad_view(self, a: Act, view: Act) -> None:
        """Give ``a`` the gradient ``view`` (a slice of another gradient buffer) without copying;
        if ``a`` already has a gradient the view is added into it."""
        g = self.grads.get(id(a))
        if g is None:
            self.grads[id(a)] = view
        elif view.b16 or g.b16:
            raise RuntimeError("hyperpri_amd: internal error: a bf16 gradient view met an earlier gradient of the same tensor")
        else:
            _lib.call("hpri_copy_slice", view.ptr, view.cs, view.coff, g.ptr, g.cs, g.coff, g.P, _rup(a.C, 4), 1, _stream())

    def param_slot(self, p: torch.Tensor) -> Tuple[torch.Tensor, int]:
        if id(p) in self.delivered:
            raise RuntimeError("hyperpri_amd: internal error: a tape node adds to a parameter gradient that an earlier segment has "
                               "already handed to autograd (the parameter is listed under the wrong stage)")
        self._touched.append(id(p))
        g = self.param_grads.get(id(p))
        if g is not None:
            return g, 1
        acc = 0
        sink = _GRAD_SINK
        if sink is not None:
            got = sink.slot(p)             # (view of a flat all-reduce bucket, accumulate?) or None
            if got is not None:
                g, acc = got
                self.param_grads[id(p)] = g
                self.sunk[id(p)] = p
                return g, int(acc)
        g = torch.empty(p.shape, dtype=torch.float32, device=p.device)
        self.param_grads[id(p)] = g
        return g, 0

    def _flush_unscale(self) -> None:
        if self._unscale:
            scale_tensors_(self._unscale, 1.0 / self.gscale)
            self._unscale.clear()

    def backward(self, lo: int = 0, hi: Optional[int] = None) -> None:
        """Run the recorded nodes ``[lo, hi)`` in reverse.  The whole tape by default; a segmented network (autograd.run_segmented:
        several chained autograd nodes sharing this tape) runs one slice per node, last slice first, and the bookkeeping is
        released with the slice that starts at 0."""
        sink = _GRAD_SINK
        hi = len(self.nodes) if hi is None else hi
        for i in range(hi - 1, lo - 1, -1):
            node, self.nodes[i] = self.nodes[i], None       # (a slice's closures go as they run)
            self._touched.clear()
            node(self)
            if sink is None or not self._touched:
                continue
            for pid in dict.fromkeys(self._touched):        # each parameter once per node, in the order the node asked
                left = self.uses.get(pid, 1) - 1            # (an op that did not announce itself counts as the only user)
                self.uses[pid] = left
                p = self.sunk.get(pid)
                if p is None or left > 0:
                    continue                                # not in a bucket, or another node still adds to this gradient
                # a bucket's all-reduce orders itself behind the CURRENT stream: before the hand-over that completes a
                # bucket, the main stream waits for the weight gradients still running on the second one
                # half-precision mode: the loss scale leaves the gradient before its bucket does.  The gradient may still be in flight
                # on the second stream, so the unscale runs where the hand-over runs: on the stream that has waited for both.
                if self.gscale != 1.0:
                    self._unscale.append(self.param_grads[pid])
                if self.used_side and getattr(sink, "completes_bucket", lambda q: True)(p):
                    # neither compute stream is held up: a third stream waits for both and hands the bucket over
                    dev = p.device
                    iss = _issue_stream(dev)
                    iss.wait_stream(torch.cuda.current_stream(dev))
                    iss.wait_stream(_side(dev))
                    with torch.cuda.stream(iss):
                        self._flush_unscale()
                        sink.ready(p)
                else:
                    if not self.used_side:
                        self._flush_unscale()
                    sink.ready(p)
        if lo > 0:
            return
        if sink is not None:
            # a parameter announced by several nodes (note_params) whose LAST announcing node never asked for its slot -- e.g. a
            # module called twice inside one tape with one result unused (that node returns before param_slot) -- still holds a
            # finished gradient in its bucket: hand it over now, or GradSync.finish() would zero it as "never landed"
            for pid, left in self.uses.items():
                p = self.sunk.get(pid)
                if p is not None and left > 0 and pid in self.param_grads:
                    if self.gscale != 1.0:
                        self._unscale.append(self.param_grads[pid])
                    if self.used_side:
                        dev = p.device
                        iss = _issue_stream(dev)
                        iss.wait_stream(torch.cuda.current_stream(dev))
                        iss.wait_stream(_side(dev))
                        with torch.cuda.stream(iss):
                            self._flush_unscale()
                            sink.ready(p)
                    else:
                        self._flush_unscale()
                        sink.ready(p)
        if self._unscale:                           # (a bucket that never completes: finish() zeroes what did not land, the rest is unscaled here)
            dev = self._unscale[0].device
            iss = _issue_stream(dev)
            iss.wait_stream(torch.cuda.current_stream(dev))
            iss.wait_stream(_side(dev))
            with torch.cuda.stream(iss):
                self._flush_unscale()
            torch.cuda.current_stream(dev).wait_stream(iss)
        self.nodes.clear()
        self.keep.clear()
        self.uses.clear()
        self.colsum.clear()
        self.bnpart.clear()
        self.gupl.clear()
        self.delivered.clear()


_IDENTITY_BN: Dict[tuple, "BNRef"] = {}


def _identity_bn(c: int, device) -> "BNRef":
    """gamma = 1, beta = 0, mean = 0, var = 1, eps = 0: an eval-mode BatchNorm that changes nothing (cached per width and device)."""
    key = (c, str(device))
    ref = _IDENTITY_BN.get(key)
    if ref is None:
        ref = BNRef.__new__(BNRef)
        ref.weight = torch.ones(c, dtype=torch.float32, device=device)
        ref.bias = torch.zeros(c, dtype=torch.float32, device=device)
        ref.running_mean = torch.zeros(c, dtype=torch.float32, device=device)
        ref.running_var = torch.ones(c, dtype=torch.float32, device=device)
        ref.num_batches_tracked = torch.zeros((), dtype=torch.long, device=device)
        ref.eps, ref.momentum = 0.0, 0.0
        _IDENTITY_BN[key] = ref
    return ref


class BNRef:
    """The tensors of one nn.BatchNorm{1,2,3}d (model_parts.py:23,26; models.py:113,172,178)."""
    __slots__ = ("weight", "bias", "running_mean", "running_var", "num_batches_tracked", "eps", "momentum")

    def __init__(self, m: torch.nn.modules.batchnorm._BatchNorm):
        if not (m.affine and m.track_running_stats):
            raise RuntimeError("hyperpri_amd: BatchNorm must be affine with running statistics (reference default)")
        self.weight, self.bias = m.weight, m.bias
        self.running_mean, self.running_var = m.running_mean, m.running_var
        self.num_batches_tracked = m.num_batches_tracked
        self.eps = float(m.eps)
        self.momentum = 0.1 if m.momentum is None else float(m.momentum)


# ---- optional per-kernel HIP-event log (bench.py's roofline leg) --------------------------------------
# When enabled, the two MFMA kernel families are bracketed by events on the launching stream and tagged
# with their template instantiation and ALGORITHMIC flops (2*M*N*K with the true, unpadded channel counts).
_EVENT_LOG = None
SHAPE_TAGS = False   # tools/layer_profile.py: key the log by problem shape too


def enable_event_log(on: bool = True):
    global _EVENT_LOG
    _EVENT_LOG = {} if on else None
    return _EVENT_LOG


class _timed:
    """``flops``: algorithmic flops of the DIRECT contraction the launch stands for (2*M*N*K, true channel counts);
    ``executed``: multiply-add flops the kernel actually issues on the matrix pipe (differs for Winograd: 16 instead of
    36 multiplies per 2x2 outputs and channel pair); defaults to ``flops``."""
    __slots__ = ("tag", "flops", "executed", "e0")

    def __init__(self, tag: str, flops: float, executed: Optional[float] = None):
        self.tag, self.flops, self.e0 = tag, flops, None
        self.executed = flops if executed is None else executed

    def __enter__(self):
        if _EVENT_LOG is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _EVENT_LOG.setdefault(self.tag, []).append((self.e0, e1, self.flops, self.executed))
        return False


def event_log_summary():
    """{tag: {"launches", "total_ms", "avg_ms", "flops_per_launch", "tflops"}} (call after a device sync)."""
    out = {}
    for tag, evs in (_EVENT_LOG or {}).items():
        ms = [a.elapsed_time(b) for a, b, _, _ in evs]
        fl = sum(f for _, _, f, _ in evs)
        ex = sum(x for _, _, _, x in evs)
        tot = sum(ms)
        out[tag] = {"launches": len(evs), "total_ms": tot, "avg_ms": tot / len(evs), "flops_per_launch": fl / len(evs),
                    "tflops": fl / (tot * 1e-3) / 1e12 if tot > 0 else 0.0,
                    "executed_flops_per_launch": ex / len(evs), "executed_tflops": ex / (tot * 1e-3) / 1e12 if tot > 0 else 0.0}
    return out


# ---- side stream: weight gradients run beside the data gradient of the same layer -------------------------
# The two GEMMs of a conv's backward are independent; issuing wgrad on a second HIP stream lets its workgroups
# fill the CUs the dgrad kernel's tail leaves idle (and vice versa).  Joined before gradients leave the node.
# Weight gradients on a second stream: a layer's weight gradient and its data gradient only share inputs, and every launch
# ends in a partial round of workgroups (4.5-18 rounds per launch at one or two workgroups per CU): the other stream's
# workgroups fill those tails.  34.35 -> 33.25 ms/step (+3.3 %), bit-identical results.  HPRI_SIDE_STREAM=0 disables it.
# Under a gradient sink (ddp.GradSync) it needs 8 hardware queues, see SIDE_STREAM_WITH_SINK below.
SIDE_STREAM = os.environ.get("HPRI_SIDE_STREAM", "1") != "0"
# (Queueing dW of layer L behind dX of layer L instead -- so that it runs beside the HBM-bound BatchNorm backward of layer L-1
# rather than beside another MFMA kernel -- measured worse: bf16 207.9 -> 204.1 cubes/s, fp32 63.42 -> 63.11;
# profiles/r04_ab_wgrad_late.txt.)
# With a gradient sink installed three or four streams are live during backward (main, weight gradients, RCCL, bucket
# hand-over).  The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a
# queue serialise: with 4 queues the second stream measured -1.6 % under a sink (one rank over RCCL), with 8 it keeps its
# +3.3 % (57.4 vs 60.4 cubes/s).  The variable is read when the runtime initialises, so it can only be raised before the
# first HIP call: done here if nothing has touched the GPU yet; otherwise the sink path stays on one stream.
def _hw_queues_ok() -> bool:
    try:
        have = int(os.environ.get("GPU_MAX_HW_QUEUES", "0"))
    except ValueError:
        have = 0
    if have >= 8:
        return True                      # set by the caller before the process started its HIP runtime (bench.py does)
    if have == 0 and not torch.cuda.is_initialized():
        os.environ["GPU_MAX_HW_QUEUES"] = "8"
        return True
    return False


_sink_env = os.environ.get("HPRI_SIDE_STREAM_SINK")
SIDE_STREAM_WITH_SINK = (_sink_env == "1") if _sink_env is not None else _hw_queues_ok()
_side_streams: Dict[int, "torch.cuda.Stream"] = {}


def _side(device) -> "torch.cuda.Stream":
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _side_streams.get(idx)
    if st is None:
        st = _side_streams[idx] = torch.cuda.Stream(device=device)
    return st


_issue_streams: Dict[int, "torch.cuda.Stream"] = {}


def _issue_stream(device) -> "torch.cuda.Stream":
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _issue_streams.get(idx)
    if st is None:
        st = _issue_streams[idx] = torch.cuda.Stream(device=device)
    return st


# Steps in flight.  A training loop that never reads a value back lets the host run many steps ahead of the GPU (a step is
# enqueued in ~5 ms and runs for 13-33 ms).  Every step in flight holds its activations, so the caching allocator keeps asking
# the driver for memory (12 allocations per step measured in a 10-step unfenced loop, 33 GiB reserved and growing) -- and the
# first process after another one has left the GPU twice spent 190 ms per step in that state.  The networks therefore wait, at
# the start of a forward, until the GPU has STARTED the previous forward (= finished the step before it): two steps in flight,
# the host still a whole step ahead, memory bounded.  HPRI_STEPS_IN_FLIGHT: 2 (default); 1 = serialised; 0 = no limit.
STEPS_IN_FLIGHT = int(os.environ.get("HPRI_STEPS_IN_FLIGHT", "2"))
_flight: Dict[int, list] = {}


def throttle(device) -> None:
    if STEPS_IN_FLIGHT <= 0 or device.type != "cuda":
        return
    idx = device.index if device.index is not None else torch.cuda.current_device()
    q = _flight.setdefault(idx, [])
    if STEPS_IN_FLIGHT == 1:                 # fully serialised: the previous step has FINISHED before this one is enqueued
        torch.cuda.current_stream(device).synchronize()
        q.clear()
        return
    # an event marks the START of a forward: with N - 1 of them outstanding, N steps are in flight once this one is enqueued
    while len(q) >= STEPS_IN_FLIGHT - 1 and q:
        q.pop(0).synchronize()
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    q.append(ev)


def join_side(device) -> None:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _side_streams.get(idx)
    if st is not None:
        torch.cuda.current_stream(device).wait_stream(st)


def _ws(nfloats: int, device) -> torch.Tensor:
    return torch.empty(max(int(nfloats), 4), dtype=torch.float32, device=device)


# ---- gradient sink (ddp.GradSync): parameter gradients are written straight into flat all-reduce buckets -------------
# sink.slot(p) -> (tensor shaped like p inside a bucket, accumulate?) or None;  sink.ready(p) is called from the tape as
# soon as p's gradient is final, so the bucket's RCCL all-reduce overlaps the rest of backward.
_GRAD_SINK = None


def set_grad_sink(sink) -> None:
    global _GRAD_SINK
    _GRAD_SINK = sink


# ---- packed-weight cache ------------------------------------------------------------------------------------------------
# The MFMA panel layouts are derived copies of the nn.Parameters.  They are rebuilt only when a parameter
# changes: the key holds the tensor's version counter (bumped by every in-place torch op: optimizer steps, copy_,
# load_state_dict), its data pointer, and _PARAM_EPOCH, which our own raw-pointer writers (FusedAdam / FusedSGD,
# synth_fill_) bump because they bypass the version counter.
_PARAM_EPOCH = 0
_BN_EPOCH = 0               # bumped whenever a training-mode forward updates running statistics (raw-pointer writes)
_PACK_CACHE: Dict[int, Dict[tuple, tuple]] = {}
# What the key CANNOT see: writes through ``p.data`` (p.data.copy_/mul_/clamp_/normal_: EMA swaps, weight clipping, .data-style
# init) leave ``p._version`` where it was.  After such a write call ``bump_param_epoch()``, or run with
#   HPRI_PACK_CACHE=0   / ``with engine.pack_cache(False):``   -- no cache: every forward repacks (~0.3 ms per C2 step), or
#   HPRI_PACK_VERIFY=1  / ``with engine.verify_packs():``      -- every cache hit is checked against a 64-bit content
#                          fingerprint of the parameter (hpri_fingerprint; one small kernel and a host read-back per hit: a
#                          debugging mode, it serialises host and device); a mismatch rebuilds the pack and counts in
#                          PACK_VERIFY_MISSES.
PACK_CACHE = os.environ.get("HPRI_PACK_CACHE", "1") != "0"
PACK_VERIFY = os.environ.get("HPRI_PACK_VERIFY", "0") == "1"
PACK_VERIFY_MISSES = 0      # stale packs the verify mode caught (a caller wrote parameter memory behind the version counter)
PACK_LAUNCHES = 0           # pack kernels launched (tests / profiles)


def bump_bn_epoch() -> None:
    """BatchNorm running statistics were rewritten (a broadcast from rank 0, a raw-pointer kernel)."""
    global _BN_EPOCH
    _BN_EPOCH += 1


def bump_param_epoch() -> None:
    """Tell the engine that parameter memory was modified behind torch's back (raw-pointer kernels, ``p.data`` writes)."""
    global _PARAM_EPOCH
    _PARAM_EPOCH += 1


class pack_cache:
    """``with engine.pack_cache(False):`` -- run the enclosed forwards without the packed-weight cache."""

    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        global PACK_CACHE
        self.old, PACK_CACHE = PACK_CACHE, self.on
        return self

    def __exit__(self, *exc):
        global PACK_CACHE
        PACK_CACHE = self.old
        return False


class verify_packs:
    """``with engine.verify_packs():`` -- check every packed-weight cache hit against the parameter's content fingerprint."""

    def __init__(self, on: bool = True):
        self.on = bool(on)

    def __enter__(self):
        global PACK_VERIFY
        self.old, PACK_VERIFY = PACK_VERIFY, self.on
        return self

    def __exit__(self, *exc):
        global PACK_VERIFY
        PACK_VERIFY = self.old
        return False


def _fingerprint(*tensors: Optional[torch.Tensor]) -> tuple:
    """Content fingerprints (host integers; synchronises) of the given device tensors."""
    live = [t for t in tensors if t is not None]
    out = torch.empty(len(live), dtype=torch.int64, device=live[0].device)
    for i, t in enumerate(live):
        _lib.call("hpri_fingerprint", _p(t), t.numel() * t.element_size() // 4, ctypes.c_void_p(out.data_ptr() + 8 * i), _stream())
    return tuple(out.tolist())


def _cached_pack(w: torch.Tensor, key: tuple, build, extra=None, also=()):
    """``extra``: further state the packed copy depends on (BN statistics of a folded conv); a mismatch rebuilds.
    ``also``: further tensors whose CONTENT the pack depends on (verify mode fingerprints them together with ``w``)."""
    global PACK_VERIFY_MISSES
    if not PACK_CACHE:
        return build()
    ent = _PACK_CACHE.get(id(w))
    try:
        ver = w._version
    except RuntimeError:        # inference tensors carry no version counter: never cache them
        return build()
    stamp = (w.data_ptr(), ver, _PARAM_EPOCH)
    key = key + (_lib.kind(),)           # (the 16-bit packs of the two libraries differ: bf16 / IEEE half)
    if ent is None or ent["stamp"] != stamp:
        if ent is None:
            import weakref
            wid = id(w)
            weakref.finalize(w, _PACK_CACHE.pop, wid, None)
        ent = _PACK_CACHE[id(w)] = {"stamp": stamp, "packs": {}}
    got = ent["packs"].get(key)
    fp = _fingerprint(w, *also) if PACK_VERIFY else None
    if got is not None and got[0] == extra and fp is not None and got[2] is not None and got[2] != fp:
        PACK_VERIFY_MISSES += 1     # same version counter, different bytes: the parameter was written through .data
        got = None
    if got is None or got[0] != extra:
        got = ent["packs"][key] = (extra, build(), fp)
    elif fp is not None and got[2] is None:
        got = ent["packs"][key] = (got[0], got[1], fp)       # first verified use of a pack built with verification off
    return got[1]


def _pack(w: torch.Tensor, mode: int, K: int, ncols: int, T: int, cup: int, d1: int) -> Tuple[torch.Tensor, int]:
    ncols_pad = _rup(ncols, 64)

    def build():
        global PACK_LAUNCHES
        n = _lib.load().hpri_packed_weight_floats(K, ncols_pad, T)
        wp = torch.empty(n, dtype=torch.float32, device=w.device)
        _lib.call("hpri_pack_weight", _p(w), _p(wp), mode, K, ncols, ncols_pad, T, cup, 0, d1, _stream())
        PACK_LAUNCHES += 1
        return wp
    return _cached_pack(w, ("f32", mode, K, ncols, T, cup, d1), build), ncols_pad


# fp32 mode: the ConvTranspose2d forms on gemm_f32v2.hip (persistent, two workgroups per CU, both operands by LDS-DMA) instead of the
# direct 1x1 kernel, where the shapes allow (Cin % 16 == 0 forward, Cup % 16 == 0 data gradient: every stage of the reference's
# networks).  A module attribute for A/B measurements, no environment switch.
GEMM_F32V2 = True
# ... its data-gradient form measured SLOWER than the direct kernel's gather on the C2 shapes (86.6 vs 95.1 TFLOP/s: 64-byte pieces of
# 16 different high-resolution pixels per DMA instruction; profiles/r04_convt_f32v2_ab.txt), the forward +7 % (97.5 vs 90.8): forward only.
GEMM_F32V2_DGRAD = False


def _pack_f32k16(w: torch.Tensor, mode: int, K: int, ncols: int, cup: int, d1: int = 0) -> Tuple[torch.Tensor, int]:
    ncols_pad = _rup(ncols, 64)

    def build():
        global PACK_LAUNCHES
        wp = torch.empty(_lib.load().hpri_packed_weight_f32k16_floats(K, ncols_pad), dtype=torch.float32, device=w.device)
        _lib.call("hpri_pack_weight_f32k16", _p(w), _p(wp), mode, K, ncols, ncols_pad, cup, d1, _stream())
        PACK_LAUNCHES += 1
        return wp
    return _cached_pack(w, ("f32k16", mode, K, ncols, cup, d1), build), ncols_pad


def _pack_bf16(w: torch.Tensor, mode: int, K: int, ncols: int, T: int, d1: int, cup: int = 0,
               split: int = 0, gap: Optional[Tuple[int, int]] = None) -> Tuple[torch.Tensor, int]:
    """``gap`` = (first channel, length) of structural-zero channels on the layer's input-channel axis (modes 0 / 1, one plane):
    K resp. ncols then count the padded axis (hpri_pack_weight_bf16_gap)."""
    ncols_pad = _rup(ncols, 64)

    def build():
        global PACK_LAUNCHES
        chunks = (K + 31) // 32
        wp = torch.empty(chunks * T * ncols_pad * 32 * (split + 1), dtype=torch.bfloat16, device=w.device)
        if gap is not None:
            _lib.call("hpri_pack_weight_bf16_gap", _p(w), _p(wp), mode, K, ncols, ncols_pad, T, d1, gap[0], gap[1], _stream())
        else:
            _lib.call("hpri_pack_weight_bf16", _p(w), _p(wp), mode, K, ncols, ncols_pad, T, d1, cup, split, _stream())
        PACK_LAUNCHES += 1
        return wp
    return _cached_pack(w, ("bf16", mode, K, ncols, T, d1, cup, split, gap), build), ncols_pad


def _conv_launch_bf16(x: Act, wp: torch.Tensor, bias: Optional[torch.Tensor], y: Act, stats: Optional[torch.Tensor],
                      N: int, H: int, W: int, cin_pad: int, cout: int, cout_pad: int, y_cw: int, ks: int,
                      amode: int = A_DIRECT, epi: int = E_DIRECT, accumulate: int = 0,
                      H2: int = 0, W2: int = 0, py0: int = 0, px0: int = 0, cup: int = 0, cin_true: int = 0,
                      split: int = 0) -> None:
    ksplit = ctypes.c_int(); tiles = ctypes.c_int(); wsf = ctypes.c_size_t()
    _lib.call("hpri_conv_fwd_bf16_plan", N, H, W, cin_pad, cout_pad, ks, amode, epi, split, ctypes.byref(ksplit),
              ctypes.byref(tiles), ctypes.byref(wsf))
    ws = _ws(wsf.value, x.buf.device) if wsf.value else None
    tag = f"conv_fwd_{('bf16', 'bf16x3', 'bf16x6')[split]}<{ks},{'2x2' if (cout_pad % 128 == 0 and split) else ('narrow' if split == 2 else '4x1')},{'s2d' if amode else 'direct'},{'d2s' if epi else 'direct'}>"
    if SHAPE_TAGS:
        tag += f" N{N} {H}x{W} K{cin_pad} N{cout}"
    with _timed(tag, 2.0 * N * H * W * (cin_true or cin_pad) * cout * ks * ks):
        _lib.call("hpri_conv_fwd_bf16", x.ptr, x.cs, x.coff, _p(wp), _p(bias), y.ptr, y.cs, y.coff, _p(stats),
                  N, H, W, cin_pad, cout, cout_pad, y_cw, ks, amode, epi, accumulate, H2, W2, py0, px0, cup, split,
                  _p(ws), wsf.value, _stream())


def _conv_launch(x: Act, wp: torch.Tensor, bias: Optional[torch.Tensor], y: Act, stats: Optional[torch.Tensor],
                 N: int, H: int, W: int, cin_pad: int, cout: int, cout_pad: int, y_cw: int, ks: int,
                 amode: int = A_DIRECT, epi: int = E_DIRECT, accumulate: int = 0,
                 H2: int = 0, W2: int = 0, py0: int = 0, px0: int = 0, cup: int = 0, cin_true: int = 0) -> None:
    ksplit = ctypes.c_int(); tiles = ctypes.c_int(); wsf = ctypes.c_size_t()
    _lib.call("hpri_conv_fwd_plan", N, H, W, cin_pad, cout_pad, ks, amode, epi, ctypes.byref(ksplit), ctypes.byref(tiles),
              ctypes.byref(wsf))
    ws = _ws(wsf.value, x.buf.device) if wsf.value else None
    tag = f"conv_fwd<{ks},{'2x2' if cout_pad % 128 == 0 else '4x1'},{'s2d' if amode else 'direct'},{'d2s' if epi else 'direct'}>"
    if SHAPE_TAGS:
        tag += f" N{N} {H}x{W} K{cin_pad} N{cout}"
    with _timed(tag, 2.0 * N * H * W * (cin_true or cin_pad) * cout * ks * ks):
        _lib.call("hpri_conv_fwd", x.ptr, x.cs, x.coff, _p(wp), _p(bias), y.ptr, y.cs, y.coff, _p(stats),
                  N, H, W, cin_pad, cout, cout_pad, y_cw, ks, amode, epi, accumulate, H2, W2, py0, px0, cup,
                  _p(ws), wsf.value, _stream())


# --------------------------------------------------------------------------------------------------
# conv (3x3 pad 1 | 1x1 | Linear) [+ BatchNorm + ReLU]
# --------------------------------------------------------------------------------------------------
# One stage = planning (which kernel family serves this layer in this precision mode), the forward launch, the BatchNorm
# finalize + apply pass, and a backward node made of three parts (BatchNorm backward, weight gradient, data gradient).  Each
# part is its own function over the stage's context ``c`` (a SimpleNamespace: what the closures of rounds 1-3 captured).
def conv_bn_relu(tape: Tape, x: Act, weight: torch.Tensor, bias: Optional[torch.Tensor], bn: Optional[BNRef],
                 train: bool, ks: int, groups: int = 1, relu: bool = True, need_dx: bool = True,
                 precision: Optional[str] = None, room: int = 0, next_cout: int = 0, cat_room: int = 0,
                 cat_into: Optional[Act] = None, k_gap: Optional[Tuple[int, int]] = None, planes_only: bool = False,
                 out_planes: bool = False, head_next: bool = False, relu_without_bn: bool = False) -> Act:
    """Conv2d(k=ks, pad=ks//2) -> BatchNorm -> ReLU  (model_parts.py:22-27; models.py:169-180 with the
    Conv3d weight (F,1,D,3,3) read as (F,D,3,3); models.py:108-114 for Linear -> BatchNorm1d -> ReLU with
    ks = 1 and ``groups`` = images, each image being its own BN batch, models.py:132).

    The last four arguments belong to the plane form of SpectralUNET's skips (``plane_gemm_mode``; 1x1 layers, bf16 mode, BN
    present): ``cat_room`` = C2 > 0: the result's planes are the first half of a padded concat [this | zeros to a multiple of
    32 | C2 channels]; ``cat_into`` = a: the result's planes are the second half of ``a``'s concat (``concat_planes`` then has
    nothing to copy); ``k_gap``: ``x`` is such a concat -- (first, length) of its structural-zero channels, which the weight does
    not have; ``planes_only``: every consumer of the result reads planes (no fp32 copy is written).

    ``head_next``: the only reader of the result is the 1x1 output layer (``out_conv``): in the bf16 mode the result is written as
    bf16 planes only and the head reads those (``head_planes_mode``)."""
    T = ks * ks
    cout = weight.shape[0]
    cin = weight.numel() // (cout * T)
    if cin + (k_gap[1] if k_gap else 0) != x.C:
        raise RuntimeError(f"hyperpri_amd: conv expects {cin} input channels, got {x.C}")
    prec = precision or DEFAULT_PRECISION
    if prec not in PRECISIONS:
        raise RuntimeError(f"hyperpri_amd: unknown precision {prec!r}; choose from {PRECISIONS}")
    if x.raw is not None:
        inner = max(next_cout, 1 if (head_next and HEAD_PLANES) else 0)
        if (bn is not None and not train and not tape.record and FOLD_EVAL_BN and prec == "bf16" and ks == 3 and groups == 1 and room == 0
                and inner > 0 and cout % 32 == 0 and PLANE_CONV and PLANES_ONLY_ACT and PLANE_PRODUCERS and not cat_room
                and cat_into is None and k_gap is None and x.H * x.W * _rup(max(cout, inner), 32) * 2 < 0x7FFFFF00):
            return _conv_ingest_eval(x, weight, bias, bn, cin, cout, relu)
        x = x.materialize()
    if bn is None and relu and relu_without_bn:
        # (``bn=None`` alone is the bare convolution / linear layer, as the kernel-level tests use it)
        # Linear / Conv -> ReLU without a BatchNorm (SpectralUNET(bnorm=False), models.py:105-110): the ReLU and its backward are the
        # BatchNorm stage's own, run with an identity "BatchNorm" in eval mode (scale 1, shift 0: y = relu(1 * x + 0) exactly; its
        # two parameter gradients land in scratch).  Until round 5 this branch returned the convolution WITHOUT the ReLU.
        bn, train = _identity_bn(cout, x.buf.device), False
    if bn is not None and not train and not tape.record and FOLD_EVAL_BN:
        return _conv_folded_eval(x, weight, bias, bn, ks, cin, cout, relu, prec, room,
                                 inner=(max(next_cout, 1 if (head_next and HEAD_PLANES) else 0) if groups == 1 else 0),
                                 out_planes=out_planes and groups == 1, cat_room=cat_room, cat_into=cat_into, k_gap=k_gap,
                                 planes_only=planes_only)
    c = types.SimpleNamespace(x=x, weight=weight, bias=bias, bn=bn, ks=ks, T=T, groups=groups, relu=relu, need_dx=need_dx, prec=prec,
                              cin=cin, cout=cout, cin_pad=x.cw, k_gap=k_gap, dev=x.buf.device, lowp=prec in LOWP,
                              split=_SPLIT.get(prec, 0), use_batch=bn is not None and train)
    _conv_pick_kernels(c, cat_room, cat_into, planes_only)
    _conv_forward(c)
    if bn is None:
        c.y, c.st = c.yr, None
    else:
        _bn_forward(c, room, next_cout, cat_room, cat_into, planes_only, out_planes, head_next)
    y = c.y
    if not tape.record:
        return y
    if bn is not None and next_cout > 0 and groups == 1 and room == 0:
        y.bn_src = (c.yr, c.st, relu)          # one consumer (the caller says so): its data-gradient kernel may do this stage's reduction
    tape.note_params(weight, bias, *((bn.weight, bn.bias) if bn is not None else ()))
    tape.nodes.append(lambda tp: _conv_backward(tp, c))
    return y


def _conv_pick_kernels(c, cat_room: int, cat_into: Optional[Act], planes_only: bool) -> None:
    """Which kernel family serves the stage: ``wino`` / ``wino_d`` (fp32 Winograd forward / data gradient), ``v2`` (bf16 planes,
    3x3: conv_bf16v3.hip), ``g3`` (bf16 planes, 1x1: gemm_bf16v3.hip); none of them: the direct kernels (fp32, or the round-1
    bf16 / bf16x3 / bf16x6 forms that convert fp32 while staging).  ``yr16``: the pre-BN tensor is stored as bf16."""
    x, bn = c.x, c.bn
    # operands by LDS-DMA from bf16 planes; the DMA offsets are 32-bit per image
    c.v2 = PLANE_CONV and c.prec == "bf16" and c.ks == 3 and _planes_fit(x, max(c.cin, c.cout))
    # 1x1 layers of the bf16 mode on planes too (gemm_bf16v3.hip): forward and data gradient
    c.g3 = PLANE_GEMM and c.prec == "bf16" and c.ks == 1 and (x.f32_valid or x.pl is not None) and _rup(x.C, 32) <= 8192
    if (cat_room or cat_into is not None or c.k_gap or planes_only) and not (c.g3 and bn is not None and PLANE_WGRAD and PLANE_PRODUCERS):
        raise RuntimeError("hyperpri_amd: internal error: the plane form of a skip concat needs the plane GEMM path (see plane_gemm_mode)")
    if not x.f32_valid and not ((c.v2 or c.g3) and x.pl is not None):
        raise RuntimeError("hyperpri_amd: internal error: a planes-only activation reached a kernel that reads fp32")
    c.wino = c.prec == "fp32" and c.ks == 3 and c.groups == 1 and _wino_ok(x, c.cout)
    c.wino_d = c.prec == "fp32" and c.ks == 3 and c.groups == 1 and _wino_ok(x, c.cin)     # the data gradient has Cin columns
    c.yr16 = False
    if c.v2 and YR_BF16 and bn is not None and c.groups == 1:
        c.yr16 = _v3_plan(x, c.cin, c.cout)[0] == 1            # (split-K problems finish in fp32: hpri_splitk_finish)
    if c.g3 and YR_BF16 and bn is not None:
        c.yr16 = True
    if c.use_batch and (x.N * x.H * x.W) // max(c.groups, 1) <= 1:
        # torch.nn.functional.batch_norm's own check (_verify_batch_size): same error, same message
        raise ValueError("Expected more than 1 value per channel when training, got input size "
                         f"torch.Size([{x.N // max(c.groups, 1)}, {c.cout}, {x.H}, {x.W}])")


def _v3_plan(x: Act, k: int, ncols: int) -> Tuple[int, int]:
    """(ksplit, statistics tiles) of the bf16 plane convolution for a K = ``k``, ``ncols``-column problem over x's pixels."""
    ksp = ctypes.c_int(); tl = ctypes.c_int(); wsf = ctypes.c_size_t()
    _lib.call("hpri_conv_bf16v3_plan", x.N, x.H, x.W, _rup(k, 32), _rup(ncols, 64), ctypes.byref(ksp), ctypes.byref(tl), ctypes.byref(wsf))
    return ksp.value, tl.value


def _conv_forward(c) -> None:
    """Pack (cached), allocate the pre-BN tensor and the statistics records, launch.  Leaves c.yr, c.stats, c.tiles, c.cout_pad."""
    x, weight, bias, dev = c.x, c.weight, c.bias, c.dev
    if c.wino:
        wp, cout_pad = _pack_wino(weight, 0, c.cin, c.cout, c.cin)
    elif c.lowp:
        wp, cout_pad = _pack_bf16(weight, 0, x.C if c.k_gap else c.cin, c.cout, c.T, c.cin, split=c.split, gap=c.k_gap)
    else:
        wp, cout_pad = _pack(weight, 0, c.cin, c.cout, c.T, 0, c.cin)
    c.cout_pad = cout_pad
    if c.yr16:
        yr = Act(torch.empty(x.P * _rup(c.cout, 8), dtype=torch.bfloat16, device=dev), x.N, x.H, x.W, c.cout, _rup(c.cout, 8), 0)
        yr.b16, yr.f32_valid = True, False
    else:
        yr = Act.new(x.N, x.H, x.W, c.cout, dev)
    stats, tiles = None, 0
    if c.use_batch:
        ksp = ctypes.c_int(); tl = ctypes.c_int(); wsf = ctypes.c_size_t()
        if c.wino:
            _lib.call("hpri_conv_wino4_plan", x.N, x.H, x.W, ctypes.byref(tl))
        elif c.v2:
            _lib.call("hpri_conv_bf16v3_plan", x.N, x.H, x.W, _rup(c.cin, 32), cout_pad, ctypes.byref(ksp), ctypes.byref(tl), ctypes.byref(wsf))
        elif c.g3:
            _lib.call("hpri_gemm_bf16v3_plan", x.N, x.H * x.W, ctypes.byref(tl))
        elif c.lowp:
            _lib.call("hpri_conv_fwd_bf16_plan", x.N, x.H, x.W, c.cin_pad, cout_pad, c.ks, A_DIRECT, E_DIRECT, c.split,
                      ctypes.byref(ksp), ctypes.byref(tl), ctypes.byref(wsf))
        else:
            _lib.call("hpri_conv_fwd_plan", x.N, x.H, x.W, c.cin_pad, cout_pad, c.ks, A_DIRECT, E_DIRECT, ctypes.byref(ksp),
                      ctypes.byref(tl), ctypes.byref(wsf))
        tiles = tl.value
        stats = torch.empty(tiles * cout_pad * 4, dtype=torch.float32, device=dev)
    if c.wino:
        _conv_launch_wino(x, wp, bias, yr, stats, c.cin, c.cout, cout_pad, yr.cw)
    elif c.v2:
        _conv_launch_v2(x, wp, bias, yr, stats, c.cin, c.cout, cout_pad, yr.cw, accumulate=4 if c.yr16 else 0)
    elif c.g3:
        _gemm_launch(x, wp, bias, yr, stats, x.C, c.cout, cout_pad, yr.cw)
    elif c.lowp:
        _conv_launch_bf16(x, wp, bias, yr, stats, x.N, x.H, x.W, c.cin_pad, c.cout, cout_pad, yr.cw, c.ks, cin_true=c.cin, split=c.split)
    else:
        _conv_launch(x, wp, bias, yr, stats, x.N, x.H, x.W, c.cin_pad, c.cout, cout_pad, yr.cw, c.ks, cin_true=c.cin)
    c.yr, c.stats, c.tiles = yr, stats, tiles


def _bn_forward(c, room: int, next_cout: int, cat_room: int, cat_into: Optional[Act], planes_only: bool, out_planes: bool,
                head_next: bool = False) -> None:
    """BatchNorm finalize (batch statistics from the conv epilogue's records, or the running ones) + normalise + ReLU.  Leaves
    c.y (the stage's output, with its bf16 planes where a plane reader follows) and c.st (mean, invstd, var, scale, shift)."""
    global _BN_EPOCH
    x, bn, yr, dev, cout = c.x, c.bn, c.yr, c.dev, c.cout
    G = c.groups if c.use_batch else 1
    st = torch.empty(5 * G * cout, dtype=torch.float32, device=dev)   # mean, invstd, var_unbiased, scale, shift
    mean, invstd, varu, scale, shift = (st[i * G * cout:(i + 1) * G * cout] for i in range(5))
    if c.use_batch:
        _BN_EPOCH += 1
        _lib.call("hpri_bn_finalize", _p(c.stats), c.tiles // G, G, c.cout_pad, cout, _p(bn.weight), _p(bn.bias),
                  bn.eps, bn.momentum, _p(mean), _p(invstd), _p(varu), _p(scale), _p(shift),
                  _p(bn.running_mean), _p(bn.running_var), _p(bn.num_batches_tracked), _stream())
    else:
        _lib.call("hpri_bn_eval_prepare", _p(bn.running_mean), _p(bn.running_var), _p(bn.weight), _p(bn.bias),
                  bn.eps, cout, _p(mean), _p(invstd), _p(scale), _p(shift), _stream())
    y = Act.new_with_room(x.N, x.H, x.W, cout, room, dev)    # room > 0: a skip tensor, written where its concat needs it
    ppg = (x.P // G)
    # bf16 plane mode: the normalise pass also writes y as bf16 planes -- what the next 3x3 convolution (and the
    # weight gradient) stage by DMA -- so no conversion pass has to read y again
    # -- for the inner tensor of a DoubleConv (``next_cout`` > 0).  The OUTPUT of a DoubleConv is read by max-pooling, the
    # transposed convolution, the concat and the 1x1 output layer, all fp32 readers: no planes for it (573 MB of writes per
    # C2 step that nobody read); ``want_pl`` still tells the pooling pass to write ITS result as planes.
    y.want_pl = 1 if ((c.v2 or c.g3) and PLANE_PRODUCERS) else 0
    # (a 1x1 layer's output is read by the next 1x1 layer or a concat in front of one: always planes)
    # (``out_planes``: the caller knows a plane reader for this output -- the next decoder stage's transposed convolution)
    ypl = new_planes(y, 1) if (y.want_pl and (next_cout > 0 or not PLANES_LAZY or c.g3 or out_planes)) else None
    if cat_room > 0:
        # first half of a padded concat: [cout | zeros to the next multiple of 32 | cat_room channels], one plane buffer
        ob = _rup(cout, 32)
        ccs = _rup(ob + cat_room, 32)
        cbuf = torch.empty(y.P * ccs, dtype=torch.bfloat16, device=dev)
        ypl = y.pl = Planes(cbuf, y.P * ccs, ccs, 0, 1, cw=ob)
        y.cat_pl = (cbuf, ccs, ob, cat_room)
    elif cat_into is not None:
        cbuf, ccs, ob, c2 = cat_into.cat_pl
        if c2 != cout or cat_into.P != y.P:
            raise RuntimeError("hyperpri_amd: internal error: concat halves do not match")
        ypl = y.pl = Planes(cbuf, y.P * ccs, ccs, ob, 1, cw=ccs - ob)
    if planes_only:
        y.f32_valid = False
    if head_next and HEAD_PLANES and c.v2 and y.want_pl and room == 0 and ypl is None and c.prec == "bf16":
        # the head is the only reader: planes for it, no fp32 copy (301 MB less written and 150 MB less read per C2 step)
        ypl = new_planes(y, 1)
        y.f32_valid = False
    cpl = None
    if ypl is None and y.want_pl and y.parent is not None and PLANES_CONCAT and y.C % 8 == 0:
        # a skip tensor: its planes go where the decoder's concat will want them -- channels [0, Cskip) of a plane buffer of the
        # concat's width; up_concat converts only the upsampled half afterwards (half the traffic of converting the concat)
        par = y.parent
        cs16 = _rup(par.C, 32)
        cpl = Planes(torch.empty(par.P * cs16, dtype=torch.bfloat16, device=dev), par.P * cs16, cs16, 0, 1)
        par.pl_part = (cpl, y.C)
        if (SKIP_PLANES_ONLY and c.v2 and c.prec == "bf16" and y.C % 32 == 0 and PLANE_CONV and PLANE_WGRAD and PLANES_ONLY_ACT
                and _planes_fit(par, par.C)):
            # the skip's readers -- max-pooling (hpri_maxpool2_*_x16), the decoder's concat (this plane buffer), the pooling
            # backward -- all read these planes: no fp32 copy of the skip is written (564 MB per C2 step) or read (2 x 564 MB)
            y.pl = Planes(cpl.buf, cpl.plane, cpl.cs, 0, 1, cw=y.C)
            y.f32_valid = False
    # ``next_cout`` > 0: y is the inner tensor of a DoubleConv (the caller says so), read only by the next 3x3 convolution
    # of ``next_cout`` columns and by that convolution's weight gradient.  When those read planes, nobody reads fp32.
    if (ypl is not None and next_cout > 0 and PLANES_ONLY_ACT and PLANE_WGRAD and room == 0
            and _planes_fit(y, max(cout, next_cout))):
        y.f32_valid = False
    _lib.call("hpri_bn_apply_relu_x16" if c.yr16 else "hpri_bn_apply_relu_pl", yr.ptr, yr.cs, yr.coff,
              y.ptr if y.f32_valid else ctypes.c_void_p(0), y.cs, y.coff,
              _p(scale), _p(shift),
              x.P, ppg, cout, y.cw, int(c.relu),
              *(_pl_args(ypl) if cpl is None else (_p(cpl.buf), cpl.plane, cpl.cs, 0, y.C, 1)), _stream())
    y.yr16 = bool(c.yr16)
    c.y, c.st = y, st


def _conv_backward(tp: Tape, c) -> None:
    g = tp.grads.pop(id(c.y), None)
    if g is None:
        return
    dyr = _conv_bwd_bn(tp, c, g)
    _conv_bwd_weight(tp, c, dyr)
    if c.need_dx:
        _conv_bwd_data(tp, c, dyr)


def _conv_bwd_bn(tp: Tape, c, g: Act) -> Act:
    """dL/d(pre-BN tensor) from dL/dy: BatchNorm (+ ReLU) backward with its parameter gradients, or -- no BatchNorm -- the bias
    gradient as a column sum.  In the plane mode the result is written as bf16 planes (and, where every reader reads planes, as
    planes only)."""
    x, bn, bias, dev, cout = c.x, c.bn, c.bias, c.dev, c.cout
    if bn is None:
        if bias is not None:
            db, acc_b = tp.param_slot(bias)
            nblk = ctypes.c_int(); cpart = ctypes.c_int()
            _lib.call("hpri_col_reduce_plan", x.P, 1, cout, ctypes.byref(nblk), ctypes.byref(cpart))
            ws = _ws(nblk.value * 2 * cpart.value + 2 * cout, dev)
            _lib.call("hpri_col_sum", g.ptr, g.cs, g.coff, _p(db), acc_b, _p(ws), ws.numel(), x.P, cout, _stream())
        return g
    # plane mode: when the weight gradient and the data gradient both read the bf16 planes, nobody reads the fp32 form
    # (1x1 layers: the plane GEMM and the plane weight gradient, gemm_bf16v3.hip / wgrad_bf16v3.hip) -- and it gets no storage
    want_dpl = (c.v2 or c.g3) and (c.need_dx or (PLANE_WGRAD and c.weight.requires_grad)) and PLANE_PRODUCERS
    f32_dead = bool(want_dpl and c.split == 0 and PLANE_WGRAD and PLANES_ONLY_GRAD and (c.need_dx or c.weight.requires_grad)
                    and ((c.ks == 3 and PLANE_CONV) or (c.g3 and _rup(cout, 32) <= 16384)))
    dyr = (Act(torch.empty(8, dtype=torch.float32, device=dev), x.N, x.H, x.W, cout, _rup(cout, 8), 0) if f32_dead
           else Act.new(x.N, x.H, x.W, cout, dev))
    G = c.groups if c.use_batch else 1
    nblk = ctypes.c_int(); cpart = ctypes.c_int()
    _lib.call("hpri_col_reduce_plan", x.P // G, G, cout, ctypes.byref(nblk), ctypes.byref(cpart))
    ws = _ws(2 * (G * nblk.value * 2 * cpart.value + G * 2 * cout), dev)
    dgam, acc_g = tp.param_slot(bn.weight)
    dbet, _ = tp.param_slot(bn.bias)
    db, acc_b = (tp.param_slot(bias) if bias is not None else (None, 0))
    st, yr = c.st, c.yr
    mean, invstd, varu, scale, shift = (st[i * G * cout:(i + 1) * G * cout] for i in range(5))
    # read by the data gradient and by the weight gradient
    dpl = new_planes(dyr, 1) if want_dpl else None
    dyr.f32_valid = not f32_dead
    bp = tp.bnpart.pop(id(c.y), None)
    _lib.call(*(("hpri_bn_relu_bwd_fused", _p(bp[0]), bp[1], bp[2]) if bp is not None else
                (("hpri_bn_relu_bwd_x16_dy16" if g.b16 else "hpri_bn_relu_bwd_x16") if c.yr16 else "hpri_bn_relu_bwd_pl",)),
              g.ptr, g.cs, g.coff, yr.ptr, yr.cs, yr.coff,
              ctypes.c_void_p(0) if f32_dead else dyr.ptr, dyr.cs, dyr.coff,
              _p(mean), _p(invstd), _p(scale), _p(shift), _p(dgam), _p(dbet), acc_g, _p(db), acc_b,
              _p(ws), ws.numel(), x.P, x.P // G, cout, dyr.cw, int(c.relu), int(c.use_batch), *_pl_args(dpl), _stream())
    return dyr


def _conv_bwd_weight(tp: Tape, c, dyr: Act) -> None:
    """dW; on the second stream when a data gradient runs beside it."""
    if not c.weight.requires_grad:          # frozen (feature_extraction, models.py:17-21): no weight gradient at all
        return
    dw, acc_w = tp.param_slot(c.weight)
    # (not for the plane GEMMs of SpectralUNET: two persistent MFMA kernels of several milliseconds side by side share one power budget
    #  and one set of CUs -- measured on one box, C3 bf16: 118.0 ms per step with the second stream, 113.8 without)
    if SIDE_STREAM and c.need_dx and not c.g3 and _EVENT_LOG is None and (_GRAD_SINK is None or SIDE_STREAM_WITH_SINK):
        main, side = torch.cuda.current_stream(c.dev), _side(c.dev)
        side.wait_stream(main)                      # dyr (and everything before it) is ready
        with torch.cuda.stream(side):
            _wgrad(c.x, dyr, dw, acc_w, c.cin, c.cout, c.ks, bf16=c.lowp, split=c.split, gap=c.k_gap)
        # what the second stream reads stays alive until the main stream has joined it at the end of backward (then it is
        # reusable at once; record_stream() instead left the blocks pending at the allocator while the host ran ahead into
        # the next step: four device allocations per bf16 step, for ever)
        tp.side_keep.extend((c.x, dyr, dw))
        tp.used_side = True
    else:
        _wgrad(c.x, dyr, dw, acc_w, c.cin, c.cout, c.ks, bf16=c.lowp, split=c.split, gap=c.k_gap)


def _conv_bwd_data(tp: Tape, c, dyr: Act) -> None:
    """dL/dx by the stage's kernel family, with the extras a consumer upstream has asked for through ``x``: column sums of the
    gradient (``colsum_req``: a ConvTranspose2d bias gradient), the upsampled half of a concat as bf16 rows (``up_slice``), the
    BatchNorm-backward sums of the stage that produced x (``bn_src``), bf16 storage for a single-reader gradient."""
    x, weight, dev, cin, cout, T = c.x, c.weight, c.dev, c.cin, c.cout, c.T
    g16 = False
    if c.v2 and GRAD_BF16_INNER and x.bn_src is not None and x.bn_src[0].b16 and tp.grads.get(id(x)) is None and x.colsum_req is None:
        g16 = _v3_plan(x, cout, cin)[0] == 1
    if g16:
        # x is the inner tensor of a DoubleConv: this launch is the only producer of dL/dx and the BatchNorm backward of
        # the stage that made x its only reader -> bf16 storage (the reader rounds its own result to bf16 anyway)
        gx = Act(torch.empty(x.P * _rup(cin, 8), dtype=torch.bfloat16, device=dev), x.N, x.H, x.W, cin, _rup(cin, 8), 0)
        gx.b16, gx.f32_valid = True, False
        tp.grads[id(x)] = gx
        acc = False
    elif c.v2 and x.skip_g16 and tp.grads.get(id(x)) is None and _y2_ok(x, cout, cin, False):
        gx, acc = None, False          # the skip half as bf16 rows, the upsampled half as planes: made in _conv_bwd_data_planes
    elif c.g3 and GRAD_BF16_GEMM and x.yr16 and x.colsum_req is None:
        # a 1x1 layer of the plane-GEMM path whose input's BatchNorm backward (or, a plane concat, both halves') reads bf16 gradients
        gx = tp.grads.get(id(x))
        acc = gx is not None
        if gx is None:
            gx = tp.grads[id(x)] = _new_grad16(x)
        elif not gx.b16:
            gx, acc = tp.grad_slot(x)
    else:
        gx, acc = tp.grad_slot(x)
    # the column sums of (a channel range of) this gradient are wanted -- the bias gradient of the ConvTranspose2d that
    # produced half of a concat: the data-gradient kernel records them per tile in its epilogue (the BatchNorm statistics
    # machinery) instead of a dedicated pass over the tensor afterwards
    gstats, gtiles = None, 0
    if COLSUM_FROM_STATS and x.colsum_req is not None and not acc and (c.wino_d or c.v2):
        if c.wino_d:
            tl = ctypes.c_int()
            _lib.call("hpri_conv_wino4_plan", x.N, x.H, x.W, ctypes.byref(tl))
            gtiles = tl.value
        else:
            gtiles = _v3_plan(x, cout, cin)[1]
        gstats = torch.empty(gtiles * _rup(cin, 64) * 4, dtype=torch.float32, device=dev)
    src = x.bn_src
    cin_cols_pad = 0
    if (c.wino_d and FUSE_BN_REDUCE and src is not None and not src[0].b16 and not acc and gstats is None
            and src[0].cs - src[0].coff >= _rup(cin, 64)):
        # this launch writes the ONLY contribution to dL/dx, and x = ReLU(BN(src[0])): its epilogue also leaves the
        # per-tile partial sums of that BatchNorm's backward (the stage that produced x then skips its reduction sweeps)
        upd, cin_cols_pad = _pack_wino(weight, 1, cout, cin, cin)
        xr, xst, xrelu = src
        tl = ctypes.c_int()
        _lib.call("hpri_conv_wino4_plan", x.N, x.H, x.W, ctypes.byref(tl))
        part = torch.empty(tl.value * 2 * cin_cols_pad, dtype=torch.float32, device=dev)
        xm, xi, _xv, xsc, xsh = (xst[i * cin:(i + 1) * cin] for i in range(5))
        wtiles = x.N * ((x.H + 1) // 2) * ((x.W + 1) // 2)
        tag = "conv_winograd_f32<3,F(2x2)>" + (f" N{x.N} {x.H}x{x.W} K{dyr.cw} N{cin}" if SHAPE_TAGS else "")
        with _timed(tag, 2.0 * x.N * x.H * x.W * cout * cin * 9, executed=2.0 * wtiles * 16 * cout * cin):
            _lib.call("hpri_conv_wino4_bnred", dyr.ptr, dyr.cs, dyr.coff, _p(upd), gx.ptr, gx.cs, gx.coff, x.N, x.H, x.W, dyr.cw,
                      cin, cin_cols_pad, gx.cw, xr.ptr, xr.cs, xr.coff, _p(xm), _p(xi), _p(xsc), _p(xsh), int(xrelu),
                      _p(part), cin_cols_pad, _stream())
        tp.bnpart[id(x)] = (part, tl.value, cin_cols_pad)
    elif c.wino_d:
        upd, cin_cols_pad = _pack_wino(weight, 1, cout, cin, cin)
        _conv_launch_wino(dyr, upd, None, gx, gstats, cout, cin, cin_cols_pad, gx.cw, accumulate=int(acc))
    elif c.v2:
        cin_cols_pad = _conv_bwd_data_planes(tp, c, dyr, gx, acc, g16, gstats)
    elif c.g3:
        wpd, cin_cols_pad = _pack_bf16(weight, 1, cout, x.C if c.k_gap else cin, T, cin, split=0, gap=c.k_gap)
        _gemm_launch(dyr, wpd, None, gx, None, cout, x.C, cin_cols_pad, gx.cw, accumulate=int(acc))
    elif c.lowp:
        wpd, cin_cols_pad = _pack_bf16(weight, 1, cout, cin, T, cin, split=c.split)
        _conv_launch_bf16(dyr, wpd, None, gx, None, x.N, x.H, x.W, dyr.cw, cin, cin_cols_pad, gx.cw, c.ks,
                          accumulate=int(acc), cin_true=cout, split=c.split)
    else:
        wpd, cin_cols_pad = _pack(weight, 1, cout, cin, T, 0, cin)
        _conv_launch(dyr, wpd, None, gx, None, x.N, x.H, x.W, dyr.cw, cin, cin_cols_pad, gx.cw, c.ks, accumulate=int(acc),
                     cin_true=cout)
    if gstats is not None:
        tp.colsum[id(x)] = (gstats, gtiles, cin_cols_pad)


def _new_grad16(x: Act) -> Act:
    """A gradient for ``x`` stored as bf16 rows (row stride = C rounded up to 8; every writer fills the pad columns with zeros)."""
    cs = _rup(x.C, 8)
    g = Act(torch.empty(x.P * cs, dtype=torch.bfloat16, device=x.buf.device), x.N, x.H, x.W, x.C, cs, 0)
    g.b16, g.f32_valid = True, False
    return g


def _y2_ok(x: Act, cout: int, cin: int, acc: bool) -> bool:
    """The data gradient of a decoder concat ``x`` can leave its upsampled half as bf16 rows (hpri_conv_bf16v3_y2)."""
    us = x.up_slice
    return bool(us is not None and CONVT_PLANES and not acc and us[0] % 64 == 0 and us[1] % 64 == 0
                and _v3_plan(x, cout, _rup(cin, 64))[0] == 1)


def _conv_bwd_data_planes(tp: Tape, c, dyr: Act, gx: Optional[Act], acc: bool, g16: bool, gstats: Optional[torch.Tensor]) -> int:
    """The 3x3 data gradient on bf16 planes (conv_bf16v3.hip, mode-1 pack); returns the padded column count of the pack."""
    x, cin, cout = c.x, c.cin, c.cout
    wpd, cin_cols_pad = _pack_bf16(c.weight, 1, cout, cin, c.T, cin, split=0)
    if g16:
        _conv_launch_v2(dyr, wpd, None, gx, None, cout, cin, cin_cols_pad, gx.cw, accumulate=4)
        return cin_cols_pad
    us = x.up_slice
    y2 = None
    if _y2_ok(x, cout, cin, acc):
        y2 = torch.empty(x.P * us[1], dtype=torch.bfloat16, device=c.dev)
    if gx is None and (y2 is None or gstats is None):
        gx, acc = tp.grad_slot(x)              # (the bf16 form of the skip half needs the second output AND the records)
    if y2 is None:
        _conv_launch_v2(dyr, wpd, None, gx, gstats, cout, cin, cin_cols_pad, gx.cw, accumulate=int(acc))
        return cin_cols_pad
    # the gradient of the upsampled half of this concat also leaves as bf16 rows (its readers, the transposed
    # convolution's data and weight gradient, stage planes); with the bias gradient coming from the statistics
    # records nobody reads that half in fp32, so it is not written
    only = gstats is not None
    flags = int(only)
    if gx is None:
        # SKIP_GRAD_BF16: the main output is the skip half alone, as bf16 rows of width Cskip
        gx = Act(torch.empty(x.P * us[0], dtype=torch.bfloat16, device=c.dev), x.N, x.H, x.W, us[0], us[0], 0)
        gx.b16, gx.f32_valid = True, False
        tp.grads[id(x)] = gx
        flags |= 2
    pl = planes_of(dyr, 1)
    with _timed("conv_planes_bf16<3,v3 256x64>" + (f" N{x.N} {x.H}x{x.W} K{_rup(cout, 32)} N{cin}" if SHAPE_TAGS else ""),
                2.0 * x.N * x.H * x.W * cout * cin * 9):
        _lib.call("hpri_conv_bf16v3_y2", _p(pl.buf), pl.cs, pl.coff, _p(wpd), ctypes.c_void_p(0), gx.ptr, gx.cs, gx.coff,
                  _p(gstats), x.N, x.H, x.W, _rup(cout, 32), cin, cin_cols_pad, us[0] if gx.b16 else gx.cw, _p(y2), us[1], 0, us[0], us[1],
                  flags, _stream())
    tp.gupl[id(x)] = (Planes(y2, x.P * us[1], us[1], 0, 1), only)
    return cin_cols_pad


# fp32 mode: the BatchNorm-backward reduction of a conv -> BN -> ReLU stage whose output has a single consumer (the inner tensor of a
# DoubleConv, CubeNET's first layer) is taken in the epilogue of that consumer's Winograd data-gradient launch
# (hpri_conv_wino4_bnred + hpri_bn_relu_bwd_fused) instead of two sweeps over the gradient and the pre-BN tensor.
# engine.FUSE_BN_REDUCE (a module attribute under the HPRI_FUSIONS master switch; the per-feature HPRI_* variables of rounds 2-3 are gone).
FUSE_BN_REDUCE = FUSIONS
# the ConvTranspose2d bias gradient from the epilogue records of the data-gradient kernel that wrote the concat's gradient
# (hpri_colsum_from_stats) instead of a pass over that tensor (hpri_col_sum).  (HPRI_FUSIONS.)
COLSUM_FROM_STATS = FUSIONS
FOLD_EVAL_BN = True   # inference only (no tape): conv + eval-mode BN + ReLU as ONE kernel with BN folded into w and b
def _bn_fold_state(bn: "BNRef", bias: Optional[torch.Tensor]):
    """What a folded pack depends on besides its weight: the versions of the BatchNorm's buffers / parameters and of the bias."""
    def ver(t):
        try:
            return (t.data_ptr(), t._version)
        except RuntimeError:
            return (t.data_ptr(), -1)
    return (_BN_EPOCH, ver(bn.running_mean), ver(bn.running_var), ver(bn.weight), ver(bn.bias), None if bias is None else ver(bias))


def _folded_pack(weight: torch.Tensor, bias: Optional[torch.Tensor], bn: BNRef, ks: int, cin: int, cout: int, prec: str, wino: bool):
    """(packed weights with the eval-mode BatchNorm's scale folded in, [scale | bias'] vector) of a folded stage; cached per weight,
    precision and BatchNorm state."""
    dev = weight.device
    T = ks * ks
    cout_pad = _rup(cout, 64)
    lowp = prec in LOWP
    split = _SPLIT.get(prec, 0)

    def build():
        global PACK_LAUNCHES
        PACK_LAUNCHES += 1
        fold = torch.empty(2 * cout, dtype=torch.float32, device=dev)
        _lib.call("hpri_bn_fold", _p(bn.running_mean), _p(bn.running_var), _p(bn.weight), _p(bn.bias), _p(bias), bn.eps, cout,
                  _p(fold[:cout]), _p(fold[cout:]), _stream())
        if wino:   # Winograd layout: the filter transform is linear, so the column scale goes in front of it
            wp = torch.empty(_lib.load().hpri_wino_packed_floats(cin, cout_pad), dtype=torch.float32, device=dev)
            _lib.call("hpri_wino4_pack", _p(weight), _p(wp), _p(fold[:cout]), 0, cin, cout, cout_pad, cin, _stream())
        elif lowp:   # bf16 / bf16x3 / bf16x6 predict path: the same fold, weights scaled in fp32 and then rounded / split
            wp = torch.empty(((cin + 31) // 32) * T * cout_pad * 32 * (split + 1), dtype=torch.bfloat16, device=dev)
            _lib.call("hpri_pack_weight_bf16_scaled", _p(weight), _p(wp), _p(fold[:cout]), cin, cout, cout_pad, T, cin, split,
                      _stream())
        else:
            wp = torch.empty(_lib.load().hpri_packed_weight_floats(cin, cout_pad, T), dtype=torch.float32, device=dev)
            _lib.call("hpri_pack_weight_scaled", _p(weight), _p(wp), _p(fold[:cout]), cin, cout, cout_pad, T, cin, _stream())
        return wp, fold

    return _cached_pack(weight, ("fold", prec, ks, "wino4" if wino else False), build, extra=_bn_fold_state(bn, bias),
                        also=(bn.running_mean, bn.running_var, bn.weight, bn.bias, bias))


def _conv_ingest_eval(x: Act, weight: torch.Tensor, bias: Optional[torch.Tensor], bn: BNRef, cin: int, cout: int, relu: bool) -> Act:
    """First-layer fused ingest of the predict path (16-bit modes; models.py:169,215-216): eval-mode Conv3x3 -> BatchNorm -> ReLU
    straight from the caller's NC(D)HW fp32 cube (``x.raw``), result as 16-bit rows = the next convolution's planes.  The same folded
    weights and the same accumulation order as ``_conv_folded_eval`` behind a layout pass: bit-identical output, without the pass."""
    global FOLD_LAUNCHES, INGEST_LAUNCHES
    FOLD_LAUNCHES += 1
    INGEST_LAUNCHES += 1
    t = x.raw[0]
    dev = t.device
    cout_pad = _rup(cout, 64)
    wp, fold = _folded_pack(weight, bias, bn, 3, cin, cout, "bf16", False)
    rows = torch.empty(x.P * cout, dtype=torch.bfloat16, device=dev)
    tag = "conv_ingest_h16<3,256x64>"
    if SHAPE_TAGS:
        tag += f" N{x.N} {x.H}x{x.W} C{cin} N{cout}"
    with _timed(tag, 2.0 * x.P * 9 * cin * cout):
        _lib.call("hpri_conv3x3_ingest_h16", _p(t), _p(wp), _p(fold[cout:]), _p(rows), cout, 0, x.N, cin, x.H, x.W, cout, cout_pad,
                  1 if relu else 0, _stream())
    y = Act(torch.empty(8, dtype=torch.float32, device=dev), x.N, x.H, x.W, cout, _rup(cout, 8), 0)
    y.f32_valid = False
    y.pl = Planes(rows, x.P * cout, cout, 0, 1)
    return y


FOLD_LAUNCHES = 0     # folded conv+BN+ReLU stages executed (tests assert that the predict path really takes them)


PREDICT_SKIP_PLANES = True     # predict path, bf16 / f16: skips and decoder concats as 16-bit planes only, transposed convolutions on the plane GEMM


PREDICT_GEMM_PLANES = True     # predict path, bf16 / f16: the 1x1 layers (SpectralUNET's Linear stack) on the plane GEMM, concats on planes


def predict_gemm_planes_ok(prec: Optional[str]) -> bool:
    return bool(PREDICT_GEMM_PLANES and FOLD_EVAL_BN and (prec or DEFAULT_PRECISION) == "bf16" and PLANE_GEMM and PLANE_PRODUCERS)


def _gemm_folded_eval(x: Act, weight: torch.Tensor, bias: Optional[torch.Tensor], bn: BNRef, cin: int, cout: int, relu: bool,
                      cat_room: int, cat_into: Optional[Act], k_gap: Optional[Tuple[int, int]], planes_only: bool) -> Act:
    """Eval-mode Linear / Conv1x1 -> BatchNorm -> ReLU of the 16-bit modes on the plane GEMM (gemm_bf16v3.hip; models.py:105-115 under
    PLTrainer.py:530-532): BatchNorm folded into the packed weights and the bias, ReLU in the epilogue, the result written as 16-bit
    rows -- into its half of a padded plane concat (``cat_room`` / ``cat_into``, as the training forward does) or into planes of its
    own; fp32 rows only when the caller has an fp32 reader (``planes_only`` False and no concat).  Until round 5 these layers ran on the
    round-1 kernel that converts fp32 activations while staging (393 TF on SpectralUNET-1650 where the plane GEMM reaches 860-1000)."""
    global FOLD_LAUNCHES, PACK_LAUNCHES
    FOLD_LAUNCHES += 1
    dev = x.buf.device
    K = x.C                                   # (a padded concat: its structural-zero channels included)
    cout_pad = _rup(cout, 64)

    def build():
        global PACK_LAUNCHES
        PACK_LAUNCHES += 1
        fold = torch.empty(2 * cout, dtype=torch.float32, device=dev)
        _lib.call("hpri_bn_fold", _p(bn.running_mean), _p(bn.running_var), _p(bn.weight), _p(bn.bias), _p(bias), bn.eps, cout,
                  _p(fold[:cout]), _p(fold[cout:]), _stream())
        # (once per weight version, cached: the column scale in fp32, then the gapped / plain pack rounds to 16 bits)
        wf = (weight.detach().reshape(cout, cin) * fold[:cout].unsqueeze(1)).contiguous()
        wp = torch.empty(((K + 31) // 32) * cout_pad * 32, dtype=torch.bfloat16, device=dev)
        if k_gap is not None:
            _lib.call("hpri_pack_weight_bf16_gap", _p(wf), _p(wp), 0, K, cout, cout_pad, 1, cin, k_gap[0], k_gap[1], _stream())
        else:
            _lib.call("hpri_pack_weight_bf16", _p(wf), _p(wp), 0, K, cout, cout_pad, 1, cin, 0, 0, _stream())
        return wp, fold

    wp, fold = _cached_pack(weight, ("fold_gemm", K, k_gap), build, extra=_bn_fold_state(bn, bias),
                            also=(bn.running_mean, bn.running_var, bn.weight, bn.bias, bias))
    fbias = fold[cout:]
    acc = 2 if relu else 0
    if cat_room == 0 and cat_into is None and not planes_only:
        y = Act.new(x.N, x.H, x.W, cout, dev)                     # an fp32 reader behind it
        _gemm_launch(x, wp, fbias, y, None, K, cout, cout_pad, y.cw, accumulate=acc)
        return y
    y = Act(torch.empty(8, dtype=torch.float32, device=dev), x.N, x.H, x.W, cout, _rup(cout, 8), 0)
    y.f32_valid = False
    if cat_room > 0:
        ob = _rup(cout, 32)
        ccs = _rup(ob + cat_room, 32)
        cbuf = torch.empty(y.P * ccs, dtype=torch.bfloat16, device=dev)
        ypl = Planes(cbuf, y.P * ccs, ccs, 0, 1, cw=ob)
        y.cat_pl = (cbuf, ccs, ob, cat_room)
    elif cat_into is not None:
        cbuf, ccs, ob, c2 = cat_into.cat_pl
        if c2 != cout or cat_into.P != y.P:
            raise RuntimeError("hyperpri_amd: internal error: concat halves do not match")
        ypl = Planes(cbuf, y.P * ccs, ccs, ob, 1, cw=ccs - ob)
    else:
        cs16 = _rup(cout, 32)
        ypl = Planes(torch.empty(y.P * cs16, dtype=torch.bfloat16, device=dev), y.P * cs16, cs16, 0, 1)
    if ypl.cw > cout_pad:
        raise RuntimeError("hyperpri_amd: internal error: plane rows wider than the packed columns")
    y.pl = ypl
    y.want_pl = 1
    rows = Act(ypl.buf, x.N, x.H, x.W, cout, ypl.cs, ypl.coff)
    rows.b16, rows.f32_valid = True, False
    _gemm_launch(x, wp, fbias, rows, None, K, cout, cout_pad, ypl.cw, accumulate=acc)      # (columns [cout, cw): exact zeros)
    return y


def _conv_folded_eval(x: Act, weight: torch.Tensor, bias: Optional[torch.Tensor], bn: BNRef, ks: int, cin: int, cout: int,
                      relu: bool, prec: str = "fp32", room: int = 0, inner: int = 0, out_planes: bool = False, cat_room: int = 0,
                      cat_into: Optional[Act] = None, k_gap: Optional[Tuple[int, int]] = None, planes_only: bool = False) -> Act:
    """Eval-mode Conv -> BatchNorm -> ReLU (running statistics) without the normalise pass: w' = w*gamma/sqrt(var+eps),
    b' = (b-mean)*gamma/sqrt(var+eps)+beta, ReLU in the conv epilogue.  Used when nothing is recorded for backward
    (torch.no_grad / inference_mode: PLTrainer.py:530,626).  ``inner`` > 0: the result is the inner tensor of a DoubleConv (``inner`` = the
    channels of the convolution that reads it), or the head's input (1); in the bf16
    mode the plane kernel then writes it as bf16 rows, which ARE the next convolution's planes (no fp32 copy, no conversion pass)."""
    global FOLD_LAUNCHES
    if ks == 1 and predict_gemm_planes_ok(prec) and room == 0 and (x.f32_valid or x.pl is not None) and _rup(x.C, 32) <= 8192:
        return _gemm_folded_eval(x, weight, bias, bn, cin, cout, relu, cat_room, cat_into, k_gap, planes_only)
    if cat_room or cat_into is not None or k_gap or planes_only:
        raise RuntimeError("hyperpri_amd: internal error: the plane form of a skip concat needs the plane GEMM path (see plane_gemm_mode)")
    FOLD_LAUNCHES += 1
    dev = x.buf.device
    T = ks * ks
    cout_pad = _rup(cout, 64)
    lowp = prec in LOWP
    split = _SPLIT.get(prec, 0)
    wino = prec == "fp32" and ks == 3 and _wino_ok(x, cout)
    if not x.f32_valid and not (lowp and PLANE_CONV and prec == "bf16" and ks == 3 and x.pl is not None and _planes_fit(x, max(cin, cout))):
        raise RuntimeError("hyperpri_amd: internal error: a planes-only activation reached a kernel that reads fp32")

    wp, fold = _folded_pack(weight, bias, bn, ks, cin, cout, prec, wino)
    fbias = fold[cout:]
    y = Act.new_with_room(x.N, x.H, x.W, cout, room, dev)
    if wino:
        _conv_launch_wino(x, wp, fbias, y, None, cin, cout, cout_pad, y.cw, accumulate=2 if relu else 0)
    elif lowp and PLANE_CONV and prec == "bf16" and ks == 3 and _planes_fit(x, max(cin, cout)):
        if (inner and room == 0 and cout % 32 == 0 and PLANES_ONLY_ACT and PLANE_PRODUCERS
                and _v3_plan(x, cin, cout)[0] == 1 and _planes_fit(y, max(cout, inner))):     # (the reader must be able to take planes too)
            # (cout a multiple of 32: the rows have no pad channels to zero; split-K problems finish in fp32)
            rows = Act(torch.empty(y.P * cout, dtype=torch.bfloat16, device=dev), x.N, x.H, x.W, cout, cout, 0)
            rows.b16 = True
            _conv_launch_v2(x, wp, fbias, rows, None, cin, cout, cout_pad, cout, accumulate=(2 if relu else 0) | 4)
            y = Act(torch.empty(8, dtype=torch.float32, device=dev), x.N, x.H, x.W, cout, _rup(cout, 8), 0)
            y.f32_valid = False
            y.pl = Planes(rows.buf, y.P * cout, cout, 0, 1)
            return y
        par = y.parent
        if (PREDICT_SKIP_PLANES and room > 0 and par is not None and SKIP_PLANES_ONLY and PLANES_CONCAT and PLANES_ONLY_ACT and PLANE_PRODUCERS
                and PLANE_WGRAD and cout % 32 == 0 and _v3_plan(x, cin, cout)[0] == 1 and _planes_fit(par, par.C)):
            # a skip tensor (round 5: as the training forward has had it since round 4): written as 16-bit rows into channels
            # [0, Cskip) of a plane buffer of the decoder concat's width and nowhere else -- max-pooling reads those rows
            # (hpri_maxpool2_fwd_x16), the decoder's transposed convolution adds its half on the plane GEMM (up_concat), the
            # convolution behind the concat stages the planes: no fp32 skip, no fp32 concat, no conversion pass
            cs16 = _rup(par.C, 32)
            cpl = Planes(torch.empty(par.P * cs16, dtype=torch.bfloat16, device=dev), par.P * cs16, cs16, 0, 1)
            par.pl_part = (cpl, cout)
            rows = Act(cpl.buf, x.N, x.H, x.W, cout, cs16, 0)
            rows.b16 = True
            _conv_launch_v2(x, wp, fbias, rows, None, cin, cout, cout_pad, cout, accumulate=(2 if relu else 0) | 4)
            y.pl = Planes(cpl.buf, cpl.plane, cpl.cs, 0, 1, cw=cout)
            y.f32_valid = False
            y.want_pl = 1
            return y
        _conv_launch_v2(x, wp, fbias, y, None, cin, cout, cout_pad, y.cw, accumulate=2 if relu else 0)
        if PREDICT_SKIP_PLANES and out_planes and room == 0 and CONVT_PLANES and PLANE_PRODUCERS and _planes_fit(y, cout):
            planes_of(y, 1)      # the next decoder stage's transposed convolution stages these (a small tensor: one conversion pass)
    elif lowp:
        _conv_launch_bf16(x, wp, fbias, y, None, x.N, x.H, x.W, x.cw, cout, cout_pad, y.cw, ks, accumulate=2 if relu else 0,
                          cin_true=cin, split=split)
    else:
        _conv_launch(x, wp, fbias, y, None, x.N, x.H, x.W, x.cw, cout, cout_pad, y.cw, ks, accumulate=2 if relu else 0,
                     cin_true=cin)
    return y


def _wgrad(x: Act, dy: Act, dw: torch.Tensor, accumulate: int, cin: int, cout: int, ks: int,
           bmode: int = A_DIRECT, dst_mode: int = 0, N: int = 0, H: int = 0, W: int = 0,
           H2: int = 0, W2: int = 0, py0: int = 0, px0: int = 0, cup: int = 0, bf16: bool = False,
           split: int = 0, gap: Optional[Tuple[int, int]] = None) -> None:
    N, H, W = (N or x.N), (H or x.H), (W or x.W)
    cin_pad = x.cw
    cout_pad = _rup(cout, 64)
    plane_route = bf16 and split == 0 and ks == 3 and bmode == A_DIRECT and dst_mode == 0 and PLANE_CONV and PLANE_WGRAD
    # 1x1 layers of the bf16 mode: both operands as planes too (wgrad_bf16v3.hip)
    plane1 = (bf16 and split == 0 and ks == 1 and bmode == A_DIRECT and dst_mode == 0 and PLANE_GEMM and PLANE_WGRAD
              and _rup(x.C, 32) <= 16384 and _rup(cout, 32) <= 16384)
    if gap is not None and not plane1:
        raise RuntimeError("hyperpri_amd: internal error: a padded concat reached a weight-gradient kernel that does not know its gap")
    if plane1:
        xpl, dpl = planes_of(x, 1), planes_of(dy, 1)
        sp = ctypes.c_int(); pcr = ctypes.c_int(); pnr = ctypes.c_int()
        npx = N * H * W
        xcw = min(xpl.cw, _rup(x.C, 32))
        _lib.call("hpri_wgrad1x1_bf16v3_plan", npx, xcw, cout_pad, ctypes.byref(sp), ctypes.byref(pcr), ctypes.byref(pnr))
        pws = _ws(sp.value * pcr.value * pnr.value, x.buf.device)
        ptag = "wgrad_planes_bf16<1,v3 256x128>"
        if SHAPE_TAGS:
            ptag += f" N{N} {H}x{W} C{xcw} N{cout}"
        with _timed(ptag, 2.0 * npx * cin * cout):
            _lib.call("hpri_wgrad1x1_bf16v3", _p(xpl.buf), xpl.cs, xpl.coff, xcw, _p(dpl.buf), dpl.cs, dpl.coff,
                      min(dpl.cw, _rup(cout, 32)), _p(pws), pws.numel(), npx, xcw, cout_pad, _stream())
        if gap is None:
            _lib.call("hpri_wgrad_reduce_ex", _p(pws), _p(dw), sp.value, pcr.value, pnr.value, cin, cout, 1, 0, 0, accumulate, _stream())
        else:
            # the slabs' columns follow the padded concat: reduce at that width, then drop the gap columns on the way into dW
            g0, gl = gap
            tmp = torch.empty((cout, x.C), dtype=torch.float32, device=x.buf.device)
            _lib.call("hpri_wgrad_reduce_ex", _p(pws), _p(tmp), sp.value, pcr.value, pnr.value, x.C, cout, 1, 0, 0, 0, _stream())
            _lib.call("hpri_copy_slice_any", _p(tmp), x.C, 0, _p(dw), cin, 0, cout, g0, 0, int(accumulate), _stream())
            _lib.call("hpri_copy_slice_any", _p(tmp), x.C, g0 + gl, _p(dw), cin, g0, cout, cin - g0, 0, int(accumulate), _stream())
        return
    if not plane_route and not (x.f32_valid and dy.f32_valid):
        # every kernel below reads fp32: a planes-only operand (bf16 plane mode) would be read as uninitialised memory
        raise RuntimeError("hyperpri_amd: internal error: a planes-only activation reached a weight-gradient kernel that reads fp32")
    splits = ctypes.c_int(); cr = ctypes.c_int(); nr = ctypes.c_int()
    _lib.call("hpri_wgrad_plan", N, H, W, cin_pad, cout_pad, ks, ctypes.byref(splits), ctypes.byref(cr), ctypes.byref(nr))
    ws = _ws(splits.value * ks * ks * cr.value * nr.value, x.buf.device)
    dy_cvalid = (4 * cup) if bmode == A_S2D else dy.cw
    tag = f"conv_wgrad{('_bf16', '_bf16x3', '_bf16x6')[split] if bf16 else ''}<{ks},{'s2d' if bmode == A_S2D else 'direct'}>"
    if SHAPE_TAGS:
        tag += f" N{N} {H}x{W} C{cin_pad} N{cout}"
    if ((not bf16) and ks == 3 and bmode == A_DIRECT and dst_mode == 0 and WINOGRAD and WINO_WGRAD and N * H * W >= 4096
            and 5 * W * max(x.cs, dy.cs) * 4 < (1 << 31)):
        # Winograd weight gradient (conv_wino.hip): 16 instead of 36 multiplies per 2x2 pixels and channel pair
        sp = ctypes.c_int(); wcr = ctypes.c_int(); wnr = ctypes.c_int()
        _lib.call("hpri_wino_wgrad_plan", N, H, W, cin_pad, cout_pad, ctypes.byref(sp), ctypes.byref(wcr), ctypes.byref(wnr))
        wws = _ws(sp.value * 16 * wcr.value * wnr.value, x.buf.device)
        wtag = "conv_wgrad_winograd_f32<3>"
        if SHAPE_TAGS:
            wtag += f" N{N} {H}x{W} C{cin_pad} N{cout}"
        with _timed(wtag, 2.0 * N * H * W * cin * cout * 9, executed=2.0 * N * ((H + 1) // 2) * ((W + 1) // 2) * 16 * cin * cout):
            _lib.call("hpri_conv_wino_wgrad", x.ptr, x.cs, x.coff, cin_pad, dy.ptr, dy.cs, dy.coff, dy.cw, _p(wws), wws.numel(),
                      N, H, W, cin_pad, cout_pad, _stream())
        _lib.call("hpri_wino_wgrad_reduce", _p(wws), _p(dw), N, H, W, cin, cin_pad, cout, cout_pad, accumulate, _stream())
        return
    if plane_route:
        # bf16 planes of both operands (written by their producers) -> LDS by DMA (conv_wgrad_bf16v2.hip)
        xpl, dpl = planes_of(x, 1), planes_of(dy, 1)
        sp = ctypes.c_int(); pcr = ctypes.c_int(); pnr = ctypes.c_int()
        _lib.call("hpri_wgrad_bf16v2_plan", N, H, W, xpl.cs, cout_pad, ctypes.byref(sp), ctypes.byref(pcr), ctypes.byref(pnr))
        pws = _ws(sp.value * 9 * pcr.value * pnr.value, x.buf.device)
        ptag = "conv_wgrad_planes_bf16<3>"
        if SHAPE_TAGS:
            ptag += f" N{N} {H}x{W} C{xpl.cs} N{cout}"
        with _timed(ptag, 2.0 * N * H * W * cin * cout * 9):
            _lib.call("hpri_conv_wgrad_bf16v2", _p(xpl.buf), xpl.cs, xpl.coff, xpl.cs - xpl.coff, _p(dpl.buf), dpl.cs, dpl.coff,
                      dpl.cs - dpl.coff, _p(pws), pws.numel(), N, H, W, xpl.cs, cout_pad, _stream())
        _lib.call("hpri_wgrad_reduce_ex", _p(pws), _p(dw), sp.value, pcr.value, pnr.value, cin, cout, 3, 0, 0, accumulate, _stream())
        return
    if bf16:
        with _timed(tag, 2.0 * N * H * W * cin * cout * ks * ks):
            _lib.call("hpri_conv_wgrad_bf16", x.ptr, x.cs, x.coff, cin_pad, dy.ptr, dy.cs, dy.coff, dy_cvalid, _p(ws), ws.numel(),
                      N, H, W, cin_pad, cout_pad, ks, bmode, H2, W2, py0, px0, cup, split, _stream())
        _lib.call("hpri_wgrad_reduce", _p(ws), _p(dw), N, H, W, cin, cin_pad, cout, cout_pad, ks, dst_mode, cup, accumulate,
                  _stream())
        return
    with _timed(tag, 2.0 * N * H * W * cin * cout * ks * ks):
        _lib.call("hpri_conv_wgrad", x.ptr, x.cs, x.coff, cin_pad, dy.ptr, dy.cs, dy.coff, dy_cvalid, _p(ws), ws.numel(),
                  N, H, W, cin_pad, cout_pad, ks, bmode, H2, W2, py0, px0, cup, _stream())
    _lib.call("hpri_wgrad_reduce", _p(ws), _p(dw), N, H, W, cin, cin_pad, cout, cout_pad, ks, dst_mode, cup, accumulate,
              _stream())


# --------------------------------------------------------------------------------------------------
# MaxPool2d(2)
# --------------------------------------------------------------------------------------------------
def maxpool2(tape: Tape, x: Act) -> Act:
    """nn.MaxPool2d(2) (floor), model_parts.py:40."""
    if x.H < 2 or x.W < 2:
        raise RuntimeError("hyperpri_amd: MaxPool2d(2) needs H, W >= 2")
    dev = x.buf.device
    x16 = not x.f32_valid                         # bf16 mode: the skip exists as planes only (SKIP_PLANES_ONLY)
    if x16 and (x.pl is None or x.pl.npl != 1):
        raise RuntimeError("hyperpri_amd: internal error: a planes-only activation without planes reached max-pooling")
    # plane mode (the input carries bf16 planes): the pooled map is written as planes too, for the next 3x3 convolution
    npl = x.pl.npl if x.pl is not None else x.want_pl
    # ... and as planes ONLY when its input was: the pooled map's readers are then the next DoubleConv's plane kernels
    only = x16 and npl == 1 and PLANE_PRODUCERS and _planes_fit(Act(x.buf, x.N, x.H // 2, x.W // 2, 2 * x.C, 2 * x.C), 2 * x.C)
    y = (Act(torch.empty(8, dtype=torch.float32, device=dev), x.N, x.H // 2, x.W // 2, x.C, _rup(x.C, 8), 0) if only
         else Act.new(x.N, x.H // 2, x.W // 2, x.C, dev))
    ypl = new_planes(y, npl) if (npl > 0 and PLANE_PRODUCERS) else None
    y.want_pl = npl
    y.f32_valid = not only
    if x16:
        _lib.call("hpri_maxpool2_fwd_x16", _p(x.pl.buf), x.pl.cs, x.pl.coff, ctypes.c_void_p(0) if only else y.ptr, y.cs, y.coff,
                  x.N, x.H, x.W, _rup(x.C, 4), *_pl_args(ypl), _stream())
    else:
        _lib.call("hpri_maxpool2_fwd_pl", x.ptr, x.cs, x.coff, y.ptr, y.cs, y.coff, x.N, x.H, x.W, x.cw, *_pl_args(ypl), _stream())
    if tape.record:
        def bwd(tp: Tape) -> None:
            g = tp.grads.pop(id(y), None)
            if g is None:
                return
            gx, acc = tp.grad_slot(x, b16_ok=True)
            if x16 or gx.b16:
                _lib.call("hpri_maxpool2_bwd_x16", _p(x.pl.buf) if x16 else x.ptr, int(x16), x.pl.cs if x16 else x.cs,
                          x.pl.coff if x16 else x.coff, g.ptr, g.cs, g.coff, gx.ptr, int(gx.b16), gx.cs, gx.coff,
                          x.N, x.H, x.W, _rup(x.C, 4), int(acc), _stream())
            else:
                _lib.call("hpri_maxpool2_bwd", x.ptr, x.cs, x.coff, g.ptr, g.cs, g.coff, gx.ptr, gx.cs, gx.coff,
                          x.N, x.H, x.W, _rup(x.C, 4), int(acc), _stream())
        tape.nodes.append(bwd)
    return y


# --------------------------------------------------------------------------------------------------
# upsample (ConvTranspose2d k2 s2 | bilinear x2) -> zero-pad -> concat with / multiply by the skip
# --------------------------------------------------------------------------------------------------
def _upsample_bwd(tp: Tape, c) -> None:
    """Backward of ``_upsample_into``: F.pad's backward (drop the ring), the ConvTranspose2d bias / weight / data gradients -- on the
    plane-fed kernels when the gradient of the upsampled half arrived as bf16 rows -- or the bilinear interpolation's adjoint."""
    dst, weight, bias, need_dx1, x1, precision, dev = c.dst, c.weight, c.bias, c.need_dx1, c.x1, c.precision, c.dev
    H2, W2, py0, px0, cup, dY, dX = c.H2, c.W2, c.py0, c.px0, c.cup, c.dY, c.dX
    gu = tp.grads.pop(id(dst), None)
    if gu is None:
        return
    if weight is None:
        if need_dx1:
            gx, acc = tp.grad_slot(x1)
            _lib.call("hpri_upsample2x_bwd", gu.ptr, gu.cs, gu.coff, gx.ptr, gx.cs, gx.coff, x1.N, x1.H, x1.W, H2, W2,
                      py0, px0, _rup(cup, 4), int(acc), _stream())
        return
    cin = weight.shape[0]
    if dY or dX:   # F.pad's backward drops the ring
        _lib.call("hpri_fill_pad", gu.ptr, gu.cs, gu.coff, gu.N, H2, W2, cup, py0, py0 + 2 * x1.H, px0, px0 + 2 * x1.W, _stream())
    cs = tp.colsum.pop(id(dst), None)
    if bias is not None:
        db, acc_b = tp.param_slot(bias)
        if cs is not None and len(cs) == 4:
            gstats, gtiles, cpad, c0 = cs
            _lib.call("hpri_colsum_from_stats", _p(gstats), gtiles, cpad, c0, cup, _p(db), acc_b, _stream())
        else:
            nblk = ctypes.c_int(); cpart = ctypes.c_int()
            _lib.call("hpri_col_reduce_plan", gu.P, 1, cup, ctypes.byref(nblk), ctypes.byref(cpart))
            ws = _ws(nblk.value * 2 * cpart.value + 2 * cup, dev)
            _lib.call("hpri_col_sum", gu.ptr, gu.cs, gu.coff, _p(db), acc_b, _p(ws), ws.numel(), gu.P, cup, _stream())
    bprec = precision or DEFAULT_PRECISION
    gp = tp.gupl.pop(id(dst), None)
    gpl = gp[0] if gp is not None else None
    if gpl is None and not gu.f32_valid:
        raise RuntimeError("hyperpri_amd: internal error: the gradient of an upsampled tensor exists as planes only, but the planes are gone")
    pw = gpl is not None and cup % 64 == 0 and _rup(cin, 32) <= 16384          # weight gradient on planes (wgrad_bf16v3.hip)
    pd = gpl is not None and cup % 32 == 0                                   # data gradient on planes (gemm_bf16v3.hip)
    if gpl is not None and not gu.f32_valid and not ((pw or not weight.requires_grad) and (pd or not need_dx1)):
        raise RuntimeError("hyperpri_amd: internal error: a planes-only gradient reached a transposed convolution that reads fp32")

    def wgrad_planes():
        xp_ = planes_of(x1, 1)
        sp = ctypes.c_int(); pcr = ctypes.c_int(); pnr = ctypes.c_int()
        xcw = min(xp_.cw, _rup(cin, 32))
        _lib.call("hpri_wgrad1x1_bf16v3_plan", x1.P, xcw, 4 * cup, ctypes.byref(sp), ctypes.byref(pcr), ctypes.byref(pnr))
        pws = _ws(sp.value * pcr.value * pnr.value, dev)
        with _timed("wgrad_planes_bf16<convT>", 2.0 * x1.P * cin * 4 * cup):
            _lib.call("hpri_wgrad_convt_bf16v3", _p(xp_.buf), xp_.cs, xp_.coff, xcw, _p(gpl.buf), gpl.cs, gpl.coff, _p(pws), pws.numel(),
                      x1.N, x1.H, x1.W, xcw, cup, H2, W2, py0, px0, _stream())
        _lib.call("hpri_wgrad_reduce_ex", _p(pws), _p(dw), sp.value, pcr.value, pnr.value, cin, 4 * cup, 1, 1, cup, acc_w, _stream())
    if weight.requires_grad and pw:
        dw, acc_w = tp.param_slot(weight)
        if SIDE_STREAM and need_dx1 and _EVENT_LOG is None and (_GRAD_SINK is None or SIDE_STREAM_WITH_SINK):
            main, side = torch.cuda.current_stream(dev), _side(dev)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                wgrad_planes()
            tp.side_keep.extend((x1, gpl, dw))
            tp.used_side = True
        else:
            wgrad_planes()
    elif weight.requires_grad:
        dw, acc_w = tp.param_slot(weight)
        if SIDE_STREAM and need_dx1 and _EVENT_LOG is None and (_GRAD_SINK is None or SIDE_STREAM_WITH_SINK):
            # the weight gradient and the data gradient of the transposed convolution only share inputs: second stream
            main, side = torch.cuda.current_stream(dev), _side(dev)
            side.wait_stream(main)                  # gu (pad ring dropped) and everything before it is ready
            with torch.cuda.stream(side):
                _wgrad(x1, gu, dw, acc_w, cin, 4 * cup, 1, bmode=A_S2D, dst_mode=1, H2=H2, W2=W2, py0=py0, px0=px0, cup=cup,
                       bf16=bprec in LOWP, split=_SPLIT.get(bprec, 0))
            tp.side_keep.extend((x1, gu, dw))
            tp.used_side = True
        else:
            _wgrad(x1, gu, dw, acc_w, cin, 4 * cup, 1, bmode=A_S2D, dst_mode=1, H2=H2, W2=W2, py0=py0, px0=px0, cup=cup,
                   bf16=bprec in LOWP, split=_SPLIT.get(bprec, 0))
    if need_dx1 and pd:
        wpd, cols_pad = _pack_bf16(weight, 3, 4 * cup, cin, 1, cup, cup, split=0)
        if GRAD_BF16_SINGLE and x1.yr16 and tp.grads.get(id(x1)) is None and cin % 8 == 0:
            # x1 (the bottleneck, or the previous decoder stage's output) has this one consumer: bf16 rows for its BatchNorm backward
            gx = Act(torch.empty(x1.P * cin, dtype=torch.bfloat16, device=dev), x1.N, x1.H, x1.W, cin, cin, 0)
            gx.b16, gx.f32_valid = True, False
            tp.grads[id(x1)] = gx
            with _timed("gemm_planes_bf16<convT,s2d>", 2.0 * x1.N * x1.H * x1.W * cin * 4 * cup):
                _lib.call("hpri_convt_dgrad_bf16v3_y16", _p(gpl.buf), gpl.cs, gpl.coff, _p(wpd), gx.ptr, gx.cs, gx.coff, x1.N, x1.H, x1.W,
                          cup, cin, cols_pad, cin, H2, W2, py0, px0, _stream())
        else:
            gx, acc = tp.grad_slot(x1)
            with _timed("gemm_planes_bf16<convT,s2d>", 2.0 * x1.N * x1.H * x1.W * cin * 4 * cup):
                _lib.call("hpri_convt_dgrad_bf16v3", _p(gpl.buf), gpl.cs, gpl.coff, _p(wpd), gx.ptr, gx.cs, gx.coff, x1.N, x1.H, x1.W, cup,
                          cin, cols_pad, gx.cw, H2, W2, py0, px0, int(acc), _stream())
    elif need_dx1:
        gx, acc = tp.grad_slot(x1)
        if bprec in LOWP:
            wpd, cols_pad = _pack_bf16(weight, 3, 4 * cup, cin, 1, cup, cup, split=_SPLIT.get(bprec, 0))
            _conv_launch_bf16(gu, wpd, None, gx, None, x1.N, x1.H, x1.W, 4 * cup, cin, cols_pad, gx.cw, 1,
                              amode=A_S2D, accumulate=int(acc), H2=H2, W2=W2, py0=py0, px0=px0, cup=cup, cin_true=4 * cup,
                              split=_SPLIT.get(bprec, 0))
        elif GEMM_F32V2_DGRAD and cup % 16 == 0 and H2 * W2 * gu.cs * 4 < 0x7FFFFF00 and x1.N * H2 * W2 < (1 << 31):
            wpd, cols_pad = _pack_f32k16(weight, 3, 4 * cup, cin, cup)
            with _timed("gemm_f32v2<convT,s2d>", 2.0 * x1.N * x1.H * x1.W * cin * 4 * cup):
                _lib.call("hpri_convt_dgrad_f32v2", gu.ptr, gu.cs, gu.coff, _p(wpd), gx.ptr, gx.cs, gx.coff, x1.N, x1.H, x1.W, cup, cin, cols_pad,
                          gx.cw, H2, W2, py0, px0, int(acc), _stream())
        else:
            wpd, cols_pad = _pack(weight, 3, 4 * cup, cin, 1, cup, cup)
            _conv_launch(gu, wpd, None, gx, None, x1.N, x1.H, x1.W, 4 * cup, cin, cols_pad, gx.cw, 1,
                         amode=A_S2D, accumulate=int(acc), H2=H2, W2=W2, py0=py0, px0=px0, cup=cup, cin_true=4 * cup)


def _upsample_into(tape: Tape, x1: Act, dst: Act, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor],
                   need_dx1: bool, precision: Optional[str] = None, dst_planes=None) -> bool:
    """Write up(x1), zero-padded to dst's H x W (left = floor(d/2), model_parts.py:73-80), into the view ``dst``.
    ``weight`` given: ConvTranspose2d(k2,s2) as one GEMM per input pixel (Cin -> 4*Cup) whose epilogue scatters the
    2x2 patches (model_parts.py:63-64); ``weight`` None: nn.Upsample(2, 'bilinear', align_corners=True) (:57)."""
    dev = x1.buf.device
    H2, W2 = dst.H, dst.W
    dY, dX = H2 - 2 * x1.H, W2 - 2 * x1.W
    py0, px0 = dY // 2, dX // 2          # Python floor division, as the reference's ``diffY // 2`` (-1 // 2 == -1)
    cup = dst.C
    if cup % 4:
        raise RuntimeError("hyperpri_amd: Up: channel counts must be multiples of 4")
    if dY < 0 or dX < 0:
        # skip smaller than the upsampled tensor: F.pad with negative widths crops (model_parts.py:77-80).  Not reached by
        # the 608 x 968 pyramids (floor pooling makes skips >= 2x), so no fused form: upsample at the natural 2H x 2W size,
        # then one shift-copy pass that crops / zero-pads each axis; backward is the same pass with the offsets negated.
        full = Act.new(x1.N, 2 * x1.H, 2 * x1.W, cup, dev)
        _upsample_into(tape, x1, full, weight, bias, need_dx1, precision)
        _lib.call("hpri_shift_copy", full.ptr, full.cs, full.coff, full.H, full.W, dst.ptr, dst.cs, dst.coff, dst.N, H2, W2,
                  py0, px0, cup, 0, _stream())
        if tape.record:
            def bwd_crop(tp: Tape) -> None:
                gu = tp.grads.pop(id(dst), None)
                if gu is None:
                    return
                gf, acc = tp.grad_slot(full)
                _lib.call("hpri_shift_copy", gu.ptr, gu.cs, gu.coff, H2, W2, gf.ptr, gf.cs, gf.coff, gf.N, full.H, full.W,
                          -py0, -px0, cup, int(acc), _stream())
            tape.nodes.append(bwd_crop)
        return False
    planes_written = False
    # the plane-fed transposed-convolution kernel (gemm_bf16v3.hip) also handles a pad ring: the ring is zeroed in the planes
    use_pl = (weight is not None and dst_planes is not None and CONVT_PLANES and x1.pl is not None and cup % 16 == 0
              and (precision or DEFAULT_PRECISION) == "bf16" and PLANES_CONVT and PLANE_CONV and PLANE_WGRAD
              and _rup(weight.shape[0], 32) <= min(x1.pl.cw, 8192))
    if (dY or dX) and not use_pl:
        _lib.call("hpri_fill_pad", dst.ptr, dst.cs, dst.coff, dst.N, H2, W2, cup, py0, py0 + 2 * x1.H, px0, px0 + 2 * x1.W, _stream())
    if weight is not None:
        cin = weight.shape[0]
        if cin != x1.C or weight.shape[1] != cup:
            raise RuntimeError("hyperpri_amd: Up: ConvTranspose2d channel mismatch")
        uprec = precision or DEFAULT_PRECISION
        bf16 = uprec in LOWP
        usplit = _SPLIT.get(uprec, 0)
        if bf16 and usplit == 0 and dst_planes is not None and (use_pl or not (dY or dX)) and PLANES_CONVT and PLANE_CONV and PLANE_WGRAD:
            # plane mode: the 2x2 patches go straight into the concat's bf16 planes (``dst_planes`` = (Planes, first channel)); the
            # fp32 form of the upsampled half has no reader (the next convolution and its weight gradient read planes)
            wp, ncols_pad = _pack_bf16(weight, 2, cin, 4 * cup, 1, cup, cup, split=0)
            pl, pc0 = dst_planes
            if use_pl:
                # the input's planes were written by its producer (``out_planes``): both operands by LDS-DMA (gemm_bf16v3.hip)
                xp_ = x1.pl
                if dY or dX:
                    # pad ring of the upsampled half inside the concat's planes: pairs of bf16 zeroed as floats (all offsets even)
                    _lib.call("hpri_fill_pad", _p(pl.buf), pl.cs // 2, pc0 // 2, dst.N, H2, W2, cup // 2, py0, py0 + 2 * x1.H, px0,
                              px0 + 2 * x1.W, _stream())
                with _timed("gemm_planes_bf16<convT,d2s+planes>", 2.0 * x1.N * x1.H * x1.W * cin * 4 * cup):
                    _lib.call("hpri_convt_fwd_bf16v3", _p(xp_.buf), xp_.cs, xp_.coff, _p(wp), _p(bias), ctypes.c_void_p(0), 0, 0,
                              _p(pl.buf), pl.cs, pc0, x1.N, x1.H, x1.W, _rup(cin, 32), cup, ncols_pad, H2, W2, py0, px0, _stream())
            else:
                with _timed("conv_fwd_bf16<1,4x1,direct,d2s+planes>", 2.0 * x1.N * x1.H * x1.W * cin * 4 * cup):
                    _lib.call("hpri_convt_fwd_bf16_pl", x1.ptr, x1.cs, x1.coff, _p(wp), _p(bias), ctypes.c_void_p(0), dst.cs, dst.coff,
                              x1.N, x1.H, x1.W, x1.cw, 4 * cup, ncols_pad, H2, W2, py0, px0, cup, _p(pl.buf), pl.cs, pc0, _stream())
            planes_written = True
        elif bf16:
            wp, ncols_pad = _pack_bf16(weight, 2, cin, 4 * cup, 1, cup, cup, split=usplit)
            _conv_launch_bf16(x1, wp, bias, dst, None, x1.N, x1.H, x1.W, x1.cw, 4 * cup, ncols_pad, 4 * cup, 1,
                              epi=E_D2S, H2=H2, W2=W2, py0=py0, px0=px0, cup=cup, cin_true=cin, split=usplit)
        elif GEMM_F32V2 and cin % 16 == 0 and x1.N * H2 * W2 < (1 << 31):
            wp, ncols_pad = _pack_f32k16(weight, 2, cin, 4 * cup, cup)
            with _timed("gemm_f32v2<convT,d2s>", 2.0 * x1.N * x1.H * x1.W * cin * 4 * cup):
                _lib.call("hpri_convt_fwd_f32v2", x1.ptr, x1.cs, x1.coff, _p(wp), _p(bias), dst.ptr, dst.cs, dst.coff, x1.N, x1.H, x1.W, cin,
                          cup, ncols_pad, H2, W2, py0, px0, _stream())
        else:
            wp, ncols_pad = _pack(weight, 2, cin, 4 * cup, 1, cup, cup)
            _conv_launch(x1, wp, bias, dst, None, x1.N, x1.H, x1.W, x1.cw, 4 * cup, ncols_pad, 4 * cup, 1,
                         epi=E_D2S, H2=H2, W2=W2, py0=py0, px0=px0, cup=cup, cin_true=cin)
        del wp
    else:
        if x1.C != cup:
            raise RuntimeError("hyperpri_amd: Up: bilinear upsampling keeps the channel count")
        _lib.call("hpri_upsample2x_fwd", x1.ptr, x1.cs, x1.coff, dst.ptr, dst.cs, dst.coff, x1.N, x1.H, x1.W, H2, W2,
                  py0, px0, cup, _stream())
    if not tape.record:
        return planes_written
    c = types.SimpleNamespace(dst=dst, weight=weight, bias=bias, need_dx1=need_dx1, x1=x1, precision=precision, dev=dev, H2=H2, W2=W2,
                              py0=py0, px0=px0, cup=cup, dY=dY, dX=dX)
    tape.note_params(weight, bias)
    tape.nodes.append(lambda tp: _upsample_bwd(tp, c))
    return planes_written


def up_concat(tape: Tape, x1: Act, skip: Act, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor],
              need_dx1: bool = True, precision: Optional[str] = None) -> Act:
    """cat([skip, pad(up(x1))], dim=1): model_parts.py:63-64,73-87 (and models.py:230-239).  One buffer
    [N,H,W,Cskip+Cup]: the skip already lives in channels [0,Cskip) when its producer allocated the buffer
    (Act.new_with_room) and is copied there otherwise; the upsampling kernel writes [Cskip,Cskip+Cup) directly; in
    backward the split is free (channel-slice views of the consumer's input gradient)."""
    dev = x1.buf.device
    cup = weight.shape[1] if weight is not None else x1.C
    if skip.N != x1.N:
        raise RuntimeError("hyperpri_amd: Up: batch mismatch")
    if skip.C % 4:
        raise RuntimeError("hyperpri_amd: Up: channel counts must be multiples of 4")
    par = skip.parent
    if par is not None and par.C == skip.C + cup and par.coff == skip.coff and par.buf is skip.buf:
        cat = par                  # the skip was produced in place (Act.new_with_room): nothing to copy
    else:
        if not skip.f32_valid:
            raise RuntimeError("hyperpri_amd: internal error: a planes-only skip tensor reached the copying form of the concat")
        cat = Act.new(skip.N, skip.H, skip.W, skip.C + cup, dev)
        _lib.call("hpri_copy_slice", skip.ptr, skip.cs, skip.coff, cat.ptr, cat.cs, cat.coff, cat.P, skip.C, 0, _stream())
    if cat.cw > cat.C:
        _lib.call("hpri_fill_pad", cat.ptr, cat.cs, cat.coff + cat.C, cat.N, cat.H, cat.W, cat.cw - cat.C, 0, 0, 0, 0, _stream())
    ups = cat.slice(skip.C, cup)
    part = cat.pl_part is not None and cat.pl_part[1] == skip.C and cat.pl is None
    # ... and only when every reader of the upsampled half reads planes: the next 3x3 convolution (PLANE_CONV) AND its weight
    # gradient (PLANE_WGRAD); with either switched off the round-1 kernels read the fp32 form, which must then exist
    direct = (part and cat.pl_part[0].cs == skip.C + cup       # no pad channels behind the concat's planes to zero
              and PLANE_CONV and PLANE_WGRAD)
    wrote = _upsample_into(tape, x1, ups, weight, bias, need_dx1, precision,
                           dst_planes=(cat.pl_part[0], skip.C) if direct else None)
    if part:
        # plane mode: the skip half of the concat's planes was written by the skip's producer; the upsampled half by the transposed
        # convolution itself, or -- pad ring, other precisions, bilinear -- by a conversion of that half only
        global PLANE_CONVERSIONS
        pl = cat.pl_part[0]
        if not skip.f32_valid:
            cat.f32_valid = False                   # the skip half exists as planes only (SKIP_PLANES_ONLY)
        if wrote:
            cat.f32_valid = False                   # channels [Cskip, Cskip + Cup) exist as planes only
        else:
            _lib.call("hpri_to_planes", ups.ptr, ups.cs, ups.coff, _p(pl.buf), pl.plane, pl.cs, skip.C, ups.P, cup, pl.cs - skip.C, 1,
                      _stream())
            PLANE_CONVERSIONS += 1
        cat.pl, cat.pl_part = pl, None
    if tape.record:
        if weight is not None and bias is not None and bias.requires_grad and cat.H == 2 * x1.H and cat.W == 2 * x1.W:
            cat.colsum_req = (skip.C, cup)     # (no pad ring: F.pad's backward would have to drop ring pixels from the sums)
        # (hpri_wgrad_convt_bf16v3 addresses the WHOLE gradient tensor with 32-bit DMA offsets and takes Cin <= 16384: beyond that --
        #  per-GPU batches of ~29 and more at 608x968 with 64 upsampled channels -- the fp32 half stays and the round-1 kernels run)
        if (weight is not None and wrote and CONVT_PLANES and cat.P * cup * 2 < 0x7FFFFF00 and _rup(weight.shape[0], 32) <= 16384):
            cat.up_slice = (skip.C, cup)       # plane mode: the consumer's data-gradient launch leaves this half's gradient as bf16 rows
            # ... and the skip half's as bf16 rows too, when nobody reads it in fp32: the skip is planes-only (its pooling backward
            # then runs on bf16), its BatchNorm backward reads bf16 gradients, and the upsampled half's fp32 form is not needed at
            # all (bias gradient from the statistics records: ``only`` in _conv_bwd_data_planes)
            cat.skip_g16 = bool(SKIP_GRAD_BF16 and not skip.f32_valid and skip.yr16 and COLSUM_FROM_STATS
                                and cat.colsum_req is not None and skip.C % 64 == 0 and cup % 64 == 0)

        def bwd(tp: Tape) -> None:
            g = tp.grads.pop(id(cat), None)
            if g is None:
                return
            if g.b16:
                # only the skip half exists, as bf16 rows of width Cskip; the upsampled half arrived as planes (tp.gupl)
                gs = Act(g.buf, g.N, g.H, g.W, skip.C, g.cs, g.coff)
                gs.b16, gs.f32_valid = True, False
                tp.set_grad_view(skip, gs)
                gu = Act(torch.empty(8, dtype=torch.float32, device=dev), g.N, g.H, g.W, cup, _rup(cup, 8), 0)
            else:
                tp.set_grad_view(skip, g.slice(0, skip.C))
                gu = g.slice(skip.C, cup)
            gp = tp.gupl.pop(id(cat), None)
            if gp is not None:
                tp.gupl[id(ups)] = gp
                gu.pl = gp[0]
                gu.f32_valid = not gp[1]
            tp.grads[id(ups)] = gu
            cs = tp.colsum.pop(id(cat), None)
            if cs is not None:
                tp.colsum[id(ups)] = cs + (skip.C,)
        tape.nodes.append(bwd)
    return cat


def up_attention(tape: Tape, x1: Act, skip: Act, weight: Optional[torch.Tensor], bias: Optional[torch.Tensor],
                 need_dx1: bool = True, precision: Optional[str] = None) -> Act:
    """skip * pad(up(x1)) -- the reference's ``use_attention`` branch, model_parts.py:84-85."""
    dev = x1.buf.device
    cup = weight.shape[1] if weight is not None else x1.C
    if cup != skip.C or skip.N != x1.N:
        raise RuntimeError("hyperpri_amd: Up(use_attention): skip and upsampled tensors must have equal shapes")
    u = Act.new(skip.N, skip.H, skip.W, cup, dev)
    if u.cw > u.C:
        _lib.call("hpri_fill_pad", u.ptr, u.cs, u.coff + u.C, u.N, u.H, u.W, u.cw - u.C, 0, 0, 0, 0, _stream())
    _upsample_into(tape, x1, u, weight, bias, need_dx1, precision)
    y = Act.new(skip.N, skip.H, skip.W, cup, dev)
    _lib.call("hpri_mul", skip.ptr, skip.cs, skip.coff, u.ptr, u.cs, u.coff, y.ptr, y.cs, y.coff, y.P, y.cw, 0, _stream())
    if tape.record:
        def bwd(tp: Tape) -> None:
            g = tp.grads.pop(id(y), None)
            if g is None:
                return
            gs, acc_s = tp.grad_slot(skip)
            _lib.call("hpri_mul", g.ptr, g.cs, g.coff, u.ptr, u.cs, u.coff, gs.ptr, gs.cs, gs.coff, y.P, _rup(cup, 4), int(acc_s), _stream())
            gu, acc_u = tp.grad_slot(u)
            _lib.call("hpri_mul", g.ptr, g.cs, g.coff, skip.ptr, skip.cs, skip.coff, gu.ptr, gu.cs, gu.coff, y.P, _rup(cup, 4), int(acc_u), _stream())
        tape.nodes.append(bwd)
    return y


def concat_channels(tape: Tape, a: Act, b: Act) -> Act:
    """torch.cat((a, b), -1) of two per-pixel feature maps (SpectralUNET skip, models.py:139-143).  When a.C is a
    multiple of 4 both halves are 16-byte aligned channel slices (vector copies, gradient views); otherwise
    (F = 1650) the second half lives at an unaligned offset and is moved by the element-granular copy."""
    dev = a.buf.device
    cat = Act.new(a.N, a.H, a.W, a.C + b.C, dev)
    aligned = a.C % 4 == 0
    if aligned:
        _lib.call("hpri_copy_slice", a.ptr, a.cs, a.coff, cat.ptr, cat.cs, cat.coff, cat.P, a.C, 0, _stream())
        _lib.call("hpri_copy_slice", b.ptr, b.cs, b.coff, cat.ptr, cat.cs, cat.coff + a.C, cat.P, _rup(b.C, 4), 0, _stream())
        tail = cat.cw - (a.C + _rup(b.C, 4))
        if tail > 0:
            _lib.call("hpri_fill_pad", cat.ptr, cat.cs, cat.coff + a.C + _rup(b.C, 4), cat.N, cat.H, cat.W, tail, 0, 0, 0, 0, _stream())
    else:
        _lib.call("hpri_copy_slice_any", a.ptr, a.cs, a.coff, cat.ptr, cat.cs, cat.coff, cat.P, a.C, 0, 0, _stream())
        _lib.call("hpri_copy_slice_any", b.ptr, b.cs, b.coff, cat.ptr, cat.cs, cat.coff + a.C, cat.P, b.C,
                  cat.cw - a.C, 0, _stream())
    if tape.record:
        def bwd(tp: Tape) -> None:
            g = tp.grads.pop(id(cat), None)
            if g is None:
                return
            # first half: aligned view (its channels beyond a.C belong to the other half -- every consumer of a
            # gradient view masks channels >= C)
            tp.set_grad_view(a, Act(g.buf, g.N, g.H, g.W, a.C, g.cs, g.coff))
            if aligned:
                tp.set_grad_view(b, g.slice(a.C, b.C))
            else:
                gb, acc = tp.grad_slot(b)
                _lib.call("hpri_copy_slice_any", g.ptr, g.cs, g.coff + a.C, gb.ptr, gb.cs, gb.coff, g.P, b.C,
                          0 if acc else gb.cw, int(acc), _stream())
        tape.nodes.append(bwd)
    return cat


def plane_gemm_mode(module, bnorm: bool = True) -> bool:
    """SpectralUNET in the bf16 mode with every switch of the plane path on: its skips are concatenated on planes (the two
    halves' producers write into one padded plane buffer) and the inner tensors exist as planes only."""
    prec = getattr(module, "hpri_precision", None) or DEFAULT_PRECISION
    return bool(bnorm and prec == "bf16" and PLANE_GEMM and PLANE_WGRAD and PLANE_PRODUCERS and PLANES_ONLY_ACT and PLANES_CAT1)


# bf16 mode, decoder stages: ConvTranspose2d forward, data gradient and weight gradient on the plane-fed kernels (gemm_bf16v3.hip,
# wgrad_bf16v3.hip); the gradient of the upsampled half of the concat arrives as bf16 rows from the data-gradient launch that
# produces it (hpri_conv_bf16v3_y2).  (HPRI_FUSIONS.)
CONVT_PLANES = FUSIONS
# bf16 mode: the gradient of the INNER tensor of a DoubleConv (one producer: the second convolution's data-gradient launch; one
# reader: the first stage's BatchNorm backward) is stored as bf16: 6 instead of 12 bytes of traffic per element.  (HPRI_FUSIONS.)
GRAD_BF16_INNER = FUSIONS


def convt_planes_mode(module) -> bool:
    prec = getattr(module, "hpri_precision", None) or DEFAULT_PRECISION
    return bool(prec == "bf16" and CONVT_PLANES and PLANE_CONV and PLANE_WGRAD and PLANE_PRODUCERS and PLANES_CONVT and PLANES_CONCAT)


# (HPRI_FUSIONS off: SpectralUNET's skips are concatenated in fp32 by copies and converted to planes afterwards.)
PLANES_CAT1 = FUSIONS
# bf16 mode: the 1x1 output layer reads the bf16 planes its producer wrote for it (hpri_outconv_*_x16) -- the last DoubleConv of the
# U-Nets, the plane concat [tail | up4] of SpectralUNET -- instead of an fp32 copy (and, SpectralUNET, instead of a copied fp32
# concat of 2 x 1650 channels: 9 of 135 ms per C3 step).  (HPRI_FUSIONS.)
HEAD_PLANES = FUSIONS
# bf16 mode: a skip tensor of the U-Nets (the output of an encoder DoubleConv, produced inside its decoder concat) exists as bf16 planes
# only, and so does the pooled map behind it; max-pooling reads and writes bf16 rows (hpri_maxpool2_fwd_x16 / _bwd_x16).  Rounding is
# monotonic, so the pooled values are bit-identical; the pooling backward picks the first maximum among the rounded values.  (HPRI_FUSIONS.)
SKIP_PLANES_ONLY = FUSIONS
# ... and the GRADIENT of such a skip is stored as bf16 rows: written by the decoder's data-gradient launch (hpri_conv_bf16v3_y2, main
# output as bf16), added to by the pooling backward (hpri_maxpool2_bwd_x16), read by the BatchNorm backward of the stage that made the
# skip (hpri_bn_relu_bwd_x16_dy16): 10 of 20 bytes per element of the four skip gradients.  (HPRI_FUSIONS.)
SKIP_GRAD_BF16 = FUSIONS
# ... as are the other single-producer, single-reader activation gradients that were still fp32: the input of every decoder stage
# (written by the transposed convolution's data gradient: hpri_convt_dgrad_bf16v3_y16) and the head's input (hpri_outconv_bwd_x16,
# dx_bf16); their one reader is the BatchNorm backward of the stage that produced the tensor.  (HPRI_FUSIONS.)
GRAD_BF16_SINGLE = FUSIONS
# SpectralUNET on the plane GEMMs: every activation gradient is stored as bf16 rows -- written by the data-gradient GEMM (bf16 view
# only), ADDED to by the second consumer's data-gradient GEMM (gemm_bf16v3's accumulate-into-bf16 form), split into channel-slice views
# for the two halves of a plane concat, read by hpri_bn_relu_bwd_x16_dy16: 6 instead of 12 bytes per element and layer.  (HPRI_FUSIONS.)
GRAD_BF16_GEMM = FUSIONS


def concat_planes(tape: Tape, a: Act, b: Act) -> Tuple[Act, Tuple[int, int]]:
    """torch.cat((a, b), -1) (models.py:139-143) when both halves already sit in one padded plane buffer (``conv_bn_relu`` with
    ``cat_room`` / ``cat_into``): nothing is copied.  Returns the concat as a planes-only Act of the PADDED width and the
    (first, length) of its structural-zero channels, which the consumer hands to ``conv_bn_relu(k_gap=...)``."""
    if a.cat_pl is None or b.pl is None or b.pl.buf is not a.cat_pl[0]:
        raise RuntimeError("hyperpri_amd: internal error: concat_planes needs halves produced with cat_room / cat_into")
    cbuf, ccs, ob, c2 = a.cat_pl
    C = ob + c2
    cat = Act(torch.empty(8, dtype=torch.float32, device=a.buf.device), a.N, a.H, a.W, C, _rup(C, 8), 0)
    cat.f32_valid = False
    cat.pl = Planes(cbuf, a.P * ccs, ccs, 0, 1)
    cat.yr16 = a.yr16 and b.yr16               # both halves' BatchNorm backward read bf16 gradients: so may the concat's be stored
    if tape.record:
        def bwd(tp: Tape) -> None:
            g = tp.grads.pop(id(cat), None)
            if g is None:
                return
            # both halves start at multiples of 32 channels of the consumer's input gradient: views (the gap columns hold exact
            # zeros: their weights are zero in the data-gradient pack)
            tp.set_grad_view(a, g.slice(0, a.C))
            tp.set_grad_view(b, g.slice(ob, b.C))
        tape.nodes.append(bwd)
    return cat, (a.C, ob - a.C)


# --------------------------------------------------------------------------------------------------
# 1x1 output conv / final Linear
# --------------------------------------------------------------------------------------------------
_PENDING = threading.local()


class _BCESlot:
    __slots__ = ("target", "used", "loss", "holder")

    def __init__(self, target: torch.Tensor):
        self.target, self.used, self.loss, self.holder = target, False, None, None


@contextlib.contextmanager
def pending_bce(target: torch.Tensor):
    """While active, the next recorded ``out_conv`` whose logits have as many elements as ``target`` (fp32, contiguous)
    also computes the mean BCE-with-logits against it; the slot says whether that happened (``used``) and holds the loss."""
    prev = getattr(_PENDING, "bce", None)
    slot = _PENDING.bce = _BCESlot(target)
    try:
        yield slot
    finally:
        _PENDING.bce = prev


def _head_weight_gapped(weight: torch.Tensor, K: int, Cw: int, gap: Tuple[int, int]) -> torch.Tensor:
    """The head's (K, Cw) weight with ``gap`` = (first, length) zero columns inserted: the layout of a padded plane concat."""
    g0, gl = gap

    def build():
        wg = torch.empty(K * (Cw + gl), dtype=torch.float32, device=weight.device)
        _lib.call("hpri_copy_slice_any", _p(weight), Cw, 0, _p(wg), Cw + gl, 0, K, g0, g0 + gl, 0, _stream())
        _lib.call("hpri_copy_slice_any", _p(weight), Cw, g0, _p(wg), Cw + gl, g0 + gl, K, Cw - g0, 0, 0, _stream())
        return wg
    return _cached_pack(weight, ("head_gap", g0, gl), build)


def out_conv(tape: Tape, x: Act, weight: torch.Tensor, bias: Optional[torch.Tensor], need_dx: bool = True, fuse_loss: bool = True,
             k_gap: Optional[Tuple[int, int]] = None):
    """nn.Conv2d(C, n_classes, 1) (model_parts.py:96) / nn.Linear(2F, n_classes) (models.py:103,143).
    Returns (logits NCHW tensor, register_grad) -- the caller hands the incoming NCHW gradient to
    ``register_grad`` before running the tape backwards.  ``fuse_loss=False``: the caller re-orders the logits afterwards,
    so a pending forward_loss() target does not line up with them element by element.

    bf16 mode: when the producer of ``x`` wrote bf16 planes for this reader (``conv_bn_relu(head_next=True)``, or the plane concat
    of SpectralUNET's last skip: ``k_gap`` = (first, length) of its structural-zero channels, which the weight does not have) the
    head reads those rows (hpri_outconv_*_x16) and no fp32 copy of ``x`` exists."""
    K = weight.shape[0]
    Cw = weight.numel() // K
    C = Cw + (k_gap[1] if k_gap else 0)
    if C != x.C:
        raise RuntimeError(f"hyperpri_amd: out conv expects {C} channels, got {x.C}")
    dev = x.buf.device
    x16 = x.pl is not None and not x.f32_valid
    if not x.f32_valid and x.pl is None:
        raise RuntimeError("hyperpri_amd: internal error: a planes-only activation without planes reached the head")
    if k_gap and not x16:
        raise RuntimeError("hyperpri_amd: internal error: a gapped concat reaches the head on planes only")
    wsrc = _head_weight_gapped(weight, K, Cw, k_gap) if k_gap else weight
    xa = (_p(x.pl.buf), x.pl.cs, x.pl.coff) if x16 else (x.ptr, x.cs, x.coff)
    y = torch.empty((x.N, K, x.H, x.W), dtype=torch.float32, device=dev)
    holder: Dict[str, torch.Tensor] = {}
    if tape.record and _lib.kind() == "f16":
        # half-precision mode: activation gradients are stored as IEEE half (5 exponent bits).  The gradient of a MEAN-reduced loss
        # w.r.t. a logit is at most 1 / (number of logits): the head multiplies what arrives by the next power of two above that
        # number, and the parameter gradients lose the factor again when they leave the tape (autograd._HipFn._backward)
        tape.gscale = float(2 ** max(0, (y.numel() - 1).bit_length()))
    slot = getattr(_PENDING, "bce", None)
    if fuse_loss and slot is not None and not slot.used and tape.record and slot.target.numel() == y.numel() and slot.target.device == dev:
        # forward_loss(): the loss of PLTrainer.py:86 computed while the logits are produced
        tgt = slot.target
        nblk = _lib.load().hpri_outconv_fwd_bce_blocks(x.N, x.H * x.W)
        part = torch.empty(nblk, dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        _lib.call("hpri_outconv_fwd_x16" if x16 else "hpri_outconv_fwd_bce", *xa, _p(wsrc), _p(bias), _p(y), _p(tgt), _p(part), nblk,
                  x.N, x.H * x.W, C, K, _stream())
        _lib.call("hpri_bce_finish", _p(part), nblk, y.numel(), _p(loss), _stream())
        # a DETACHED alias of the logits (same storage, same version counter, no grad_fn): holding ``y`` itself would close the
        # cycle ctx -> holder -> y -> grad_fn -> ctx and keep a never-backpropagated tape alive until the cyclic collector runs
        holder["bce_y"], holder["bce_t"] = y.detach(), tgt
        holder["bce_ver"] = (y._version, tgt._version)
        slot.used, slot.loss, slot.holder = True, loss, holder
    elif x16:
        _lib.call("hpri_outconv_fwd_x16", *xa, _p(wsrc), _p(bias), _p(y), ctypes.c_void_p(0), ctypes.c_void_p(0), 0,
                  x.N, x.H * x.W, C, K, _stream())
    else:
        _lib.call("hpri_outconv_fwd", x.ptr, x.cs, x.coff, _p(weight), _p(bias), _p(y), x.N, x.H * x.W, C, K, _stream())
    if tape.record:
        def bwd(tp: Tape) -> None:
            gy = holder.pop("g", None)
            if gy is None:
                return
            gs = holder.pop("bce_g", None)              # the scalar gradient arriving at the fused loss (None: loss unused)
            marker = holder.pop("bce_marker", None)
            logits, tgt = holder.pop("bce_y", None), holder.pop("bce_t", None)
            ver = holder.pop("bce_ver", None)
            if gs is not None and ver is not None and (logits._version, tgt._version) != ver:
                # what save_for_backward would have caught (nn.BCEWithLogitsLoss saves both): the kernels below re-read them
                raise RuntimeError("hyperpri_amd: one of the variables needed for gradient computation has been modified by an inplace "
                                   "operation: the logits or the target of forward_loss() changed between forward and backward")
            fused = gs is not None and marker is not None and gy.data_ptr() == marker.data_ptr() and not any(gy.stride())
            if gs is not None and not fused:
                # the logits had another consumer besides the fused loss: autograd has summed its gradient with the
                # (all-zero) marker, so the loss's own share is added here by the stand-alone kernel
                g2 = torch.empty_like(logits)
                _lib.call("hpri_bce_logits_bwd", _p(logits), _p(tgt), logits.numel(), _p(gs), _p(g2), _stream())
                gy = gy + g2
            gy = logits if fused else gy.contiguous()
            if tp.gscale != 1.0:
                if fused:
                    _lib.call("hpri_set_loss_scale", tp.gscale)       # (the fused heads multiply it into the gradient they form)
                else:
                    gy = gy.clone()                                   # (autograd's tensor is not ours to scale in place)
                    scale_tensors_([gy], tp.gscale)
            dw, acc_w = tp.param_slot(weight)
            db = None
            if bias is not None:
                db, _ = tp.param_slot(bias)
            nblk = ctypes.c_int(); cpart = ctypes.c_int()
            _lib.call("hpri_outconv_bwd_plan", x.N, x.H * x.W, C, K, ctypes.byref(nblk), ctypes.byref(cpart))
            ws = _ws(nblk.value * K * 2 * cpart.value, dev)
            g16 = 0
            if (need_dx and x16 and K == 1 and x.yr16 and tp.grads.get(id(x)) is None
                    and (GRAD_BF16_GEMM if k_gap else GRAD_BF16_SINGLE)):
                # the head is the first writer of this gradient and its readers read bf16: bf16 rows for the last BatchNorm
                # backward (U-Nets), or for the views of SpectralUNET's last plane concat (tail's half is added to later)
                gx = tp.grads[id(x)] = _new_grad16(x)
                gxp, gcs, gco, gcw, acc, g16 = gx.ptr, gx.cs, 0, gx.cw, False, 1
            elif need_dx:
                gx, acc = tp.grad_slot(x)
                gxp, gcs, gco, gcw = gx.ptr, gx.cs, gx.coff, gx.cw
            else:
                gxp, gcs, gco, gcw, acc = ctypes.c_void_p(0), 0, 0, 0, False
            # a gapped weight: its gradient lands in a (K, C) scratch row and moves to the parameter's layout afterwards
            dwk = torch.empty(K * C, dtype=torch.float32, device=dev) if k_gap else dw
            dbk = torch.empty(K, dtype=torch.float32, device=dev) if (k_gap and db is not None) else db
            acc_k = 0 if k_gap else acc_w
            if x16:
                _lib.call("hpri_outconv_bwd_x16", _p(gy), _p(tgt) if fused else ctypes.c_void_p(0), _p(gs) if fused else ctypes.c_void_p(0),
                          *xa, _p(wsrc), gxp, g16, gcs, gco, gcw, int(acc), _p(dwk), _p(dbk), acc_k, _p(ws), ws.numel(),
                          x.N, x.H * x.W, C, K, _stream())
            elif fused:
                _lib.call("hpri_outconv_bwd_bce", _p(gy), _p(tgt), _p(gs), x.ptr, x.cs, x.coff, _p(weight), gxp, gcs, gco, gcw,
                          int(acc), _p(dw), _p(db), acc_w, _p(ws), ws.numel(), x.N, x.H * x.W, C, K, _stream())
            else:
                _lib.call("hpri_outconv_bwd", _p(gy), x.ptr, x.cs, x.coff, _p(weight), gxp, gcs, gco, gcw, int(acc),
                          _p(dw), _p(db), acc_w, _p(ws), ws.numel(), x.N, x.H * x.W, C, K, _stream())
            if tp.gscale != 1.0 and fused:
                _lib.call("hpri_set_loss_scale", 1.0)
            if k_gap:
                g0, gl = k_gap
                _lib.call("hpri_copy_slice_any", _p(dwk), C, 0, _p(dw), Cw, 0, K, g0, 0, acc_w, _stream())
                _lib.call("hpri_copy_slice_any", _p(dwk), C, g0 + gl, _p(dw), Cw, g0, K, Cw - g0, 0, acc_w, _stream())
                if db is not None:
                    _lib.call("hpri_copy_slice_any", _p(dbk), 1, 0, _p(db), 1, 0, K, 1, 0, acc_w, _stream())
        tape.note_params(weight, bias)
        tape.nodes.append(bwd)
    return y, holder


# --------------------------------------------------------------------------------------------------
# synthetic data on the device (bench / tests): same counter-based generator as synth.py
# --------------------------------------------------------------------------------------------------
def synth_fill_(t: torch.Tensor, seed: int, mode: int = 0, thr: float = 0.0, scale: float = 1.0) -> torch.Tensor:
    _require_cuda(t, "synth target")
    if not t.is_contiguous():
        raise RuntimeError("synth_fill_: tensor must be contiguous")
    _lib.call("hpri_synth_fill", _p(t), t.numel(), seed % (1 << 64), mode, thr, scale, _stream())
    bump_param_epoch()          # t may be parameter storage
    return t
