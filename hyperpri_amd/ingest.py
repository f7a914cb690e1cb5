"""Ingest fast path for hyperspectral cubes -- opt-in, ``src/dataset.py`` stays untouched.

``HyperpriDataset.__getitem__`` (dataset.py:261-271) loads an ENVI cube as (H, W, B) float32, moves the band axis to
the front on the host (a strided 560 MB copy), slices ``[hsi_lo:hsi_hi]`` and unsqueezes for CubeNET; the default
collate stacks and Lightning copies the batch over PCIe from pageable memory; the network then needs channels-last
again.  ``CubeStager`` takes the (H, W, B) arrays as they are:

    stager = CubeStager(batch=2, height=608, width=968, bands=299, hsi_lo=0, hsi_hi=238, device="cuda:0")
    for k, item in enumerate(loader_of_hwb_arrays):
        np.copyto(stager.host_slot(k % 2), item)          # or read the ENVI file straight into the pinned slot
        ...
        x = stager.submit()                               # (N,1,238,H,W) view, H2D + layout on a side stream
        pred = cubenet(x)                                 # zero-copy: no layout kernel, no extra HBM pass

* pinned host slots (``slots`` deep) so the H2D copy of batch k+1 overlaps the step on batch k;
* fp32 sources: the H2D copy itself lands band-sliced in the zero-padded channels-last layout (one 2-D memcpy, no
  kernel at all);  fp16 sources (half the PCIe bytes; quantises reflectance to 11 bits -- NOT the reference's
  numerics, opt-in): linear H2D + one convert/pad pass;
* the tensor handed back has the reference's logical shape -- (N,1,C,H,W) for CubeNET (``unsqueeze_hsi``,
  dataset.py:269-270), (N,C,H,W) for SpectralUNET -- and is an ordinary strided view, usable by any torch op; the
  hyperpri_amd modules recognise it and consume the buffer in place.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import numpy as np
import torch

from . import _lib
from .engine import _p, _rup


class CubeStager:
    def __init__(self, batch: int, height: int, width: int, bands: int, hsi_lo: int = 0, hsi_hi: Optional[int] = None,
                 device="cuda:0", unsqueeze_hsi: bool = True, src_dtype=np.float32, slots: int = 2,
                 direct_h2d: bool = True):
        hsi_hi = bands if hsi_hi is None else hsi_hi
        if not (0 <= hsi_lo < hsi_hi <= bands):
            raise ValueError(f"CubeStager: bad band range [{hsi_lo}:{hsi_hi}] of {bands}")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("hyperpri_amd: CubeStager needs a ROCm device; there is no CPU fallback")
        self.N, self.H, self.W, self.B = batch, height, width, bands
        self.lo, self.C = hsi_lo, hsi_hi - hsi_lo
        self.cs = _rup(self.C, 8)
        self.unsqueeze = unsqueeze_hsi
        self.src_dtype = np.dtype(src_dtype)
        if self.src_dtype not in (np.dtype(np.float32), np.dtype(np.float16)):
            raise ValueError("CubeStager: src_dtype must be float32 or float16")
        self.direct = bool(direct_h2d) and self.src_dtype == np.dtype(np.float32)
        tdt = torch.float32 if self.src_dtype == np.dtype(np.float32) else torch.float16
        self.slots = slots
        self._host = [torch.empty((batch, height, width, bands), dtype=tdt).pin_memory() for _ in range(slots)]
        self._raw = None if self.direct else [torch.empty((batch, height, width, bands), dtype=tdt, device=self.device)
                                               for _ in range(slots)]
        # pad channels are zeroed once here; neither the 2-D copy nor the kernel's valid range ever dirties them
        self._dev = [torch.zeros((batch, height, width, self.cs), dtype=torch.float32, device=self.device)
                     for _ in range(slots)]
        self._stream = torch.cuda.Stream(device=self.device)
        # the zero fill above runs on the caller's stream: the copy stream must not overtake it (it would be zeroed
        # AFTER the first cube landed), and the caching allocator must know both streams use these buffers
        self._stream.wait_stream(torch.cuda.current_stream(self.device))
        for t in self._dev + (self._raw or []):
            t.record_stream(self._stream)
        self._ready: List[Optional[torch.cuda.Event]] = [None] * slots
        self._consumed: List[Optional[torch.cuda.Event]] = [None] * slots
        self._next = 0

    def host_slot(self, slot: Optional[int] = None) -> np.ndarray:
        """The pinned (N,H,W,B) staging array of the slot the next ``submit()`` will send -- fill it in place.
        Blocks until that slot's previous host-to-device copy has finished reading it (every other wait of the stager is
        GPU-side, so without this a host loop that never synchronises could overwrite a slot whose DMA is still queued)."""
        k = self._next if slot is None else slot
        if self._ready[k] is not None:
            self._ready[k].synchronize()
        return self._host[k].numpy()

    def submit(self) -> torch.Tensor:
        """Send the current slot; returns the device view.  The caller's current stream waits for the transfer; the
        slot's device buffer is reused ``slots`` submits later (the stager waits for the consumer's work first)."""
        k = self._next
        self._next = (k + 1) % self.slots
        cur = torch.cuda.current_stream(self.device)
        done = self._consumed[k]
        with torch.cuda.device(self.device), torch.cuda.stream(self._stream):
            if done is None and self._ready[k] is not None:
                done = torch.cuda.Event()                # release() was not called: wait for everything enqueued so far
                done.record(cur)
            if done is not None:
                self._stream.wait_event(done)            # work that read this slot's previous contents has finished
            self._consumed[k] = None
            P = self.N * self.H * self.W
            s = ctypes.c_void_p(self._stream.cuda_stream)
            if self.direct:
                _lib.call("hpri_hwb_h2d", ctypes.c_void_p(self._host[k].data_ptr()), _p(self._dev[k]), P, self.B, self.lo,
                          self.C, self.cs, s)
            else:
                self._raw[k].copy_(self._host[k], non_blocking=True)
                _lib.call("hpri_hwb_ingest", _p(self._raw[k]), 0 if self.src_dtype == np.dtype(np.float32) else 1,
                          _p(self._dev[k]), P, self.B, self.lo, self.C, self.cs, self.cs, s)
            ev = torch.cuda.Event()
            ev.record(self._stream)
        self._ready[k] = ev
        cur.wait_event(ev)
        x = self._dev[k][..., :self.C].permute(0, 3, 1, 2)      # logical (N,C,H,W), channels-last strides
        if self.unsqueeze:
            x = x.unsqueeze(1)                                   # (N,1,C,H,W) as dataset.py:269-270
        x._hpri_zero_padded = True
        x._hpri_slot = k
        return x

    def release(self, slot_tensor=None) -> None:
        """Mark a slot as consumed up to this point of the current stream (call it after the step that used it has been
        enqueued, e.g. after ``loss.backward()``).  ``slot_tensor`` is the tensor ``submit()`` returned (or its slot
        index); without it the most recently submitted slot is meant."""
        if slot_tensor is None:
            k = (self._next - 1) % self.slots
        elif isinstance(slot_tensor, int):
            k = slot_tensor
        else:
            k = getattr(slot_tensor, "_hpri_slot", None)
            if k is None:
                raise ValueError("CubeStager.release: not a tensor returned by submit()")
        if not 0 <= k < self.slots:
            raise ValueError(f"CubeStager.release: slot {k} out of range")
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._consumed[k] = ev


def from_hwb(cube: torch.Tensor, hsi_lo: int = 0, hsi_hi: Optional[int] = None, unsqueeze_hsi: bool = True) -> torch.Tensor:
    """Device-resident (N,H,W,B) fp32/fp16 cube -> the zero-copy network input (one slice/convert/pad pass)."""
    if cube.device.type != "cuda" or cube.dim() != 4 or cube.dtype not in (torch.float32, torch.float16):
        raise RuntimeError("hyperpri_amd: from_hwb needs a (N,H,W,B) fp32/fp16 tensor on a ROCm device")
    cube = cube.contiguous()
    N, H, W, B = cube.shape
    hsi_hi = B if hsi_hi is None else hsi_hi
    if not (0 <= hsi_lo < hsi_hi <= B):
        raise ValueError(f"from_hwb: bad band range [{hsi_lo}:{hsi_hi}] of {B}")
    C = hsi_hi - hsi_lo
    cs = _rup(C, 8)
    dst = torch.empty((N, H, W, cs), dtype=torch.float32, device=cube.device)
    with torch.cuda.device(cube.device):
        _lib.call("hpri_hwb_ingest", _p(cube), 0 if cube.dtype == torch.float32 else 1, _p(dst), N * H * W, B, hsi_lo, C, cs,
                  cs, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    x = dst[..., :C].permute(0, 3, 1, 2)
    if unsqueeze_hsi:
        x = x.unsqueeze(1)
    x._hpri_zero_padded = True
    return x
