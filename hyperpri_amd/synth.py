"""Counter-based synthetic data for HyperPRI cubes, masks and weights.

u(seed, idx) = (splitmix64_mix((seed * 0x9E3779B97F4A7C15 + idx) mod 2^64) >> 40) / 2^24
gives an exact fp32 in [0, 1) that depends only on (seed, idx), so the CPU side (numpy, here)
and the device side (``hpri_synth_uniform`` in csrc/elementwise.hip) regenerate identical
tensors without shipping 560 MB cubes.

Shapes follow the reference's data contract: RGB (N,3,H,W) ``dataset.py:257``, HSI for
SpectralUNET (N,D,H,W), HSI for CubeNET (N,1,D,H,W) ``dataset.py:269-271``; masks are
float32 (N,1,H,W) ``dataset.py:294-295``.
"""
from __future__ import annotations

import math

import numpy as np

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

SEED_CUBE = 1234   # x[n] uses seed SEED_CUBE + n
SEED_MASK = 4321   # mask[n] uses seed SEED_MASK + n
SEED_PARAM = 1000  # k-th parameter (registration order) uses seed SEED_PARAM + k


def _mix(z: np.ndarray) -> np.ndarray:
    z = z ^ (z >> np.uint64(30))
    z = z * _M1
    z = z ^ (z >> np.uint64(27))
    z = z * _M2
    z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed: int, count: int, offset: int = 0) -> np.ndarray:
    """float32 array u(seed, offset + i), i in [0, count)."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed % (1 << 64)) * GOLDEN
        out = np.empty(count, dtype=np.float32)
        step = 1 << 22
        for lo in range(0, count, step):
            hi = min(count, lo + step)
            idx = np.arange(offset + lo, offset + hi, dtype=np.uint64)
            z = _mix(base + idx)
            out[lo:hi] = (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
    return out


def image_batch(shape, seed0: int = SEED_CUBE) -> np.ndarray:
    """Batch of images/cubes: sample n is uniform(seed0 + n) over the per-sample linear index."""
    n = shape[0]
    per = int(np.prod(shape[1:]))
    out = np.empty(shape, dtype=np.float32)
    for i in range(n):
        out[i] = uniform(seed0 + i, per).reshape(shape[1:])
    return out


def mask_batch(n: int, h: int, w: int, seed0: int = SEED_MASK, thresh: float = 0.9) -> np.ndarray:
    """Throughput mask: ~10 % positives, float32 (N,1,H,W)."""
    out = np.empty((n, 1, h, w), dtype=np.float32)
    for i in range(n):
        out[i, 0] = (uniform(seed0 + i, h * w) > np.float32(thresh)).reshape(h, w)
    return out


def polyline_mask(n: int, h: int, w: int, seed0: int = SEED_MASK) -> np.ndarray:
    """Root-like mask for Dice runs: a few 1-3 px wide random-walk polylines per image
    (README.md:21 describes thin roots); deterministic in (seed0 + n)."""
    out = np.zeros((n, 1, h, w), dtype=np.float32)
    for i in range(n):
        r = uniform(seed0 + i, 4096)
        k = 0
        nroots = 3 + int(r[k] * 4); k += 1
        for _ in range(nroots):
            x = r[k] * (w - 1); k += 1
            y = 0.0
            width = 1 + int(r[k] * 3); k += 1
            drift = (r[k] - 0.5) * 1.5; k += 1
            while y < h and k < 4090:
                xi, yi = int(x), int(y)
                out[i, 0, yi:yi + 1, max(0, xi):min(w, xi + width)] = 1.0
                y += 1.0
                if (yi & 7) == 0:
                    drift += (r[k] - 0.5) * 0.8; k += 1
                    drift = max(-1.5, min(1.5, drift))
                x = min(max(x + drift, 0.0), w - 1.0)
    return out


def fan_in_of(shape) -> int:
    """fan_in as torch.nn.init._calculate_fan_in_and_fan_out computes it (dim1 x receptive field)."""
    if len(shape) < 2:
        raise ValueError("fan_in undefined for 1-d tensors")
    rf = 1
    for s in shape[2:]:
        rf *= int(s)
    return int(shape[1]) * rf


def param_values(k: int, shape, fan_in: int) -> np.ndarray:
    """w_k = (2 u(SEED_PARAM + k, .) - 1) / sqrt(fan_in)  (PyTorch-default bound 1/sqrt(fan_in))."""
    cnt = int(np.prod(shape))
    u = uniform(SEED_PARAM + k, cnt)
    return ((2.0 * u - 1.0) * np.float32(1.0 / math.sqrt(fan_in))).astype(np.float32).reshape(shape)
