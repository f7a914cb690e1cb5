"""Item queues of the persistent MFMA kernels (include/hyperpri_hip.h: hpri_set_item_queue; csrc/common.h) through the C ABI:
with a queue registered for the stream the workgroups of hpri_conv_bf16v3 / hpri_gemm_bf16v3 / hpri_convt_fwd_f32v2 DRAW their
work items instead of walking fixed lists.  Results are bit-identical to the fixed lists (same items, same arithmetic), every
launch zeroes the half of the counters the next launch on the stream will use, every item is computed exactly once also when a second kernel holds
compute units while the launch runs (tools/cu_hog.hip: workgroups that become resident late find the queues empty), and the
K-sliced (split-K) form and the channel-block-major item order take the same path.  Needs a real MI355X: ``-m gpu``."""
import ctypes
import os
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rup(x, m):
    return (x + m - 1) // m * m


def P(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


@pytest.fixture(scope="module")
def lib():
    from hyperpri_amd import _lib
    return _lib.load()


class _Queue:
    """A caller-owned queue on a stream of its own (so that nothing else in the test process shares its registration)."""

    def __init__(self, lib):
        self.lib = lib
        self.stream = torch.cuda.Stream(device=DEV)
        self.h = ctypes.c_void_p(self.stream.cuda_stream)
        self.buf = torch.zeros(lib.hpri_item_queue_bytes() // 4, dtype=torch.int32, device=DEV)
        torch.cuda.synchronize()

    def on(self):
        assert self.lib.hpri_set_item_queue(P(self.buf), self.buf.numel() * 4, self.h) == 0, self.lib.hpri_last_error()
        self.launches = 0                   # (registration resets the parity: the first launch draws from half 0)

    def next_half_is_zero(self, taken=True):
        """After a launch: the half the NEXT launch will draw from has been zeroed, the one just used holds its tickets.
        ``taken=False``: a launch whose workgroups have one item each keeps its fixed lists and leaves the queue alone."""
        if not taken:
            return int(self.buf.abs().sum()) == 0
        self.launches += 1
        h = self.buf.numel() // 2
        nxt = self.buf[:h] if self.launches % 2 == 0 else self.buf[h:]
        used = self.buf[h:] if self.launches % 2 == 0 else self.buf[:h]
        return int(nxt.abs().sum()) == 0 and int(used.abs().sum()) > 0

    def off(self):
        assert self.lib.hpri_set_item_queue(ctypes.c_void_p(0), 0, self.h) == 0


@pytest.fixture
def queue(lib):
    q = _Queue(lib)
    yield q
    torch.cuda.synchronize()
    q.off()


def _conv_problem(lib, N, H, W, K, Cols, seed):
    torch.manual_seed(seed)
    cs16, cols_pad, cw = rup(K, 32), rup(Cols, 64), rup(Cols, 8)
    npx = N * H * W
    planes = torch.zeros(npx, cs16, dtype=torch.bfloat16, device=DEV)
    planes[:, :K] = torch.randn(npx, K, device=DEV).to(torch.bfloat16)
    w = torch.randn(Cols, K, 3, 3, device=DEV) * 0.05
    b = torch.randn(Cols, device=DEV)
    wp = torch.empty((cs16 // 32) * 9 * cols_pad * 32, dtype=torch.bfloat16, device=DEV)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.hpri_pack_weight_bf16(P(w), P(wp), 0, K, Cols, cols_pad, 9, K, 0, 0, st) == 0, lib.hpri_last_error()
    k, tl, wsf = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
    assert lib.hpri_conv_bf16v3_plan(N, H, W, cs16, cols_pad, ctypes.byref(k), ctypes.byref(tl), ctypes.byref(wsf)) == 0
    torch.cuda.synchronize()
    return dict(planes=planes, wp=wp, b=b, cs16=cs16, cols_pad=cols_pad, cw=cw, npx=npx, ksplit=k.value, tiles=tl.value, ws_floats=wsf.value,
                N=N, H=H, W=W, Cols=Cols)


def _conv_run(lib, pr, q):
    y = torch.full((pr["npx"], pr["cw"]), float("nan"), device=DEV)
    stats = torch.full((pr["tiles"] * pr["cols_pad"] * 4,), float("nan"), device=DEV)
    ws = torch.empty(max(pr["ws_floats"], 4), device=DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(q.stream):
        rc = lib.hpri_conv_bf16v3(P(pr["planes"]), 0, pr["cs16"], 0, P(pr["wp"]), P(pr["b"]), P(y), pr["cw"], 0, P(stats), pr["N"], pr["H"],
                                  pr["W"], pr["cs16"], pr["Cols"], pr["cols_pad"], pr["cw"], 0, 0, P(ws), ws.numel(), q.h)
    assert rc == 0, lib.hpri_last_error()
    torch.cuda.synchronize()
    return y, stats


#                                   N  H    W    K    Cols         what the shape exercises
@pytest.mark.parametrize("shape", [(2, 304, 484, 64, 128),       # 2300 items: 4-5 per workgroup, XCD bands
                                   (2, 304, 484, 64, 512),       # eight channel blocks: channel-block-major item order
                                   (2, 76, 121, 96, 512),        # ... with 72 items per band for 64 workgroups: fixed lists
                                   (2, 38, 60, 512, 256),        # a split-K plan: fixed lists
                                   (1, 19, 30, 40, 64)])         # fewer items than workgroups
def test_conv_bf16v3_queue_equals_fixed_lists(lib, queue, shape):
    pr = _conv_problem(lib, *shape, seed=5)
    queue.off()
    y0, s0 = _conv_run(lib, pr, queue)
    assert torch.isfinite(y0[:, :pr["Cols"]]).all()
    queue.on()
    for rep in range(3):                         # the kernel re-arms the counters itself: launch after launch on one queue
        y1, s1 = _conv_run(lib, pr, queue)
        assert torch.equal(y0, y1), (shape, rep)
        if pr["ksplit"] == 1:
            assert torch.equal(s0, s1), (shape, rep)
        # (launches with fewer than two items per workgroup keep their fixed lists and leave the queue alone)
        assert queue.next_half_is_zero(taken=shape[1] >= 304), (shape, rep)


@pytest.fixture(scope="module")
def hog():
    src, so = os.path.join(ROOT, "tools", "cu_hog.hip"), os.path.join(ROOT, "tools", "bin", "libcuhog.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so, src])
    h = ctypes.CDLL(so)
    h.cu_hog_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
    return h


def test_workgroups_that_start_late_find_the_queue_empty(lib, queue, hog):
    """64 compute units are held by sleeping workgroups (64 KB of LDS each: only one of the convolution's two workgroups fits
    beside one) for the whole launch: some of the 2 x CUs workgroups become resident only when others retire.  Every item is still
    computed once (outputs and statistics equal to the undisturbed launch), and the next launch's counters are zeroed."""
    pr = _conv_problem(lib, 2, 304, 484, 64, 128, seed=6)
    queue.off()
    y0, s0 = _conv_run(lib, pr, queue)
    queue.on()
    hs = torch.cuda.Stream(device=DEV)
    sink = torch.zeros(256, dtype=torch.int32, device=DEV)
    for rep in range(2):
        torch.cuda.synchronize()
        assert hog.cu_hog_launch(64, 40.0, sink.data_ptr(), hs.cuda_stream) == 0       # 40 ms, bounded by its own deadline
        y1, s1 = _conv_run(lib, pr, queue)
        assert torch.equal(y0, y1) and torch.equal(s0, s1)
        assert queue.next_half_is_zero()


def test_gemm_bf16v3_queue_equals_fixed_lists(lib, queue):
    N, HW, K, C = 2, 100100, 160, 200          # 2 x 392 tiles x 2 column blocks: 196 items per band for 64 workgroups
    torch.manual_seed(7)
    kp, cp, cw = rup(K, 32), rup(C, 64), rup(C, 4)
    xp = torch.zeros(N * HW, kp, dtype=torch.bfloat16, device=DEV)
    xp[:, :K] = torch.randn(N * HW, K, device=DEV).to(torch.bfloat16)
    w = torch.randn(C, K, device=DEV) * 0.1
    b = torch.randn(C, device=DEV)
    wp = torch.empty((kp // 32) * cp * 32, dtype=torch.bfloat16, device=DEV)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.hpri_pack_weight_bf16(P(w), P(wp), 0, K, C, cp, 1, K, 0, 0, st) == 0, lib.hpri_last_error()
    tl = ctypes.c_int()
    assert lib.hpri_gemm_bf16v3_plan(N, HW, ctypes.byref(tl)) == 0
    torch.cuda.synchronize()

    def run():
        y = torch.full((N * HW, cw), float("nan"), device=DEV)
        stats = torch.full((tl.value * cp * 4,), float("nan"), device=DEV)
        torch.cuda.synchronize()
        with torch.cuda.stream(queue.stream):
            rc = lib.hpri_gemm_bf16v3(P(xp), kp, 0, P(wp), P(b), P(y), cw, 0, P(None), 0, 0, P(stats), cp, N, HW, kp, C, cp, cw, 0, queue.h)
        assert rc == 0, lib.hpri_last_error()
        torch.cuda.synchronize()
        return y, stats
    queue.off()
    y0, s0 = run()
    ref = xp[:, :K].double().cpu() @ w.to(torch.bfloat16).double().cpu().T + b.double().cpu()
    assert float((y0[:, :C].double().cpu() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    queue.on()
    for rep in range(3):
        y1, s1 = run()
        assert torch.equal(y0, y1) and torch.equal(s0, s1)
        assert queue.next_half_is_zero()


def test_convt_fwd_f32v2_queue_equals_fixed_lists(lib, queue):
    """The fp32 ConvTranspose2d(k=2, s=2) forward GEMM (gemm_f32v2.hip, depth-to-space epilogue) through the engine's own call."""
    import hyperpri_amd as HP
    from hyperpri_amd import engine
    torch.manual_seed(8)
    up = HP.Up(128, 64, bilinear=False).to(DEV).train()
    x1 = torch.randn(2, 128, 304, 484, device=DEV)        # 1150 pixel tiles x 2 column blocks: 287 items per band
    x2 = torch.randn(2, 64, 608, 968, device=DEV)
    sd = {k: v.clone() for k, v in up.state_dict().items()}

    def run(on):
        up.load_state_dict(sd)
        old = engine.ITEM_QUEUE
        engine.ITEM_QUEUE = on
        try:
            with torch.cuda.stream(queue.stream):
                if not on:
                    queue.off()
                y = up(x1, x2)
            torch.cuda.synchronize()
        finally:
            engine.ITEM_QUEUE = old
        return y.detach().clone()
    torch.cuda.synchronize()
    y0 = run(False)
    engine._item_queues.pop((0, queue.stream.cuda_stream), None)
    y1 = run(True)                              # (the engine registers its own queue for this stream)
    assert (0, queue.stream.cuda_stream) in engine._item_queues
    assert torch.equal(y0, y1)
    eq = engine._item_queues[(0, queue.stream.cuda_stream)]
    halves = (int(eq[:eq.numel() // 2].abs().sum()), int(eq[eq.numel() // 2:].abs().sum()))
    assert min(halves) == 0 and max(halves) > 0, halves          # (the transposed convolution drew from one half and zeroed the other)
