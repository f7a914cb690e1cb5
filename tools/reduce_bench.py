#!/usr/bin/env python3
"""Times of the weight-gradient slab reductions on the CubeNET-64 shapes (HIP events, one stream):
ConvTranspose2d reduce (hpri_wgrad_reduce, dst mode 1), Winograd reduce (hpri_wino_wgrad_reduce), BatchNorm finalize."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyperpri_amd import _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, H, W, cin, cup) in [(2, 38, 60, 1024, 512), (2, 76, 121, 512, 256), (2, 152, 242, 256, 128), (2, 304, 484, 128, 64)]:
    sp = ctypes.c_int(); cr = ctypes.c_int(); nr = ctypes.c_int()
    cout = 4 * cup
    lib.hpri_wgrad_plan(N, H, W, cin, cout, 1, ctypes.byref(sp), ctypes.byref(cr), ctypes.byref(nr))
    ws = torch.randn(sp.value * cr.value * nr.value, device=dev)
    dw = torch.zeros(cin, cup, 2, 2, device=dev)
    t = timeit(lambda: lib.hpri_wgrad_reduce(P(ws), P(dw), N, H, W, cin, cin, cout, cout, 1, 1, cup, 0, st))
    # check against torch: dW[c][co][tap] = sum_k ws[k][tap*cup+co][c]
    ref = ws.view(sp.value, nr.value, cr.value)[:, :cout, :cin].sum(0).view(4, cup, cin).permute(2, 1, 0).reshape(cin, cup, 2, 2)
    err = float((dw - ref).abs().max())
    print(f"convT reduce {cin}->{cup}: splits {sp.value} slab {ws.numel()*4/2**20:.0f} MB  {t:.1f} us  max err {err:.2e}")
