#!/usr/bin/env python3
"""The reference's own N > 1 caller on one rank: the benched step (CubeNET-64, two 238x608x968 cubes) wrapped in STOCK
torch.nn.parallel.DistributedDataParallel over RCCL (world 1: the collective moves nothing, the reducer's bookkeeping, bucket copies
and the chain of autograd nodes are all there) against the plain loop, fp32 and bf16, arms interleaved.  What Lightning
strategy="ddp" (PLTrainer.py:434-442) costs this path per step before any wire time.
usage: ddp_stock_bench.py > profiles/r05_ddp_stock_one_rank.json"""
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29577"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from torch.nn.parallel import DistributedDataParallel as DDP  # noqa: E402

import bench  # noqa: E402
import hyperpri_amd as HP  # noqa: E402
from hyperpri_amd import autograd, engine  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
saved = os.dup(1); os.dup2(2, 1)          # (librccl's banner goes to fd 1)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier(); torch.cuda.synchronize()
sys.stdout.flush(); os.dup2(saved, 1); os.close(saved)
out = {"what": "CubeNET-64, 2 x 238x608x968, forward + BCE + backward; plain loop (one autograd node: SEGMENT_AUTO off) vs stock DDP over RCCL, "
               "world 1 (chain of autograd nodes, 25 MiB buckets, broadcast_buffers=False)", "library_stamp": bench._lib_stamp(), "modes": {}}
for prec in ("fp32", "bf16", "f16"):
    net = HP.CubeNET(238, 1, first_depth=64, bilinear=False).to(dev).train()
    bench.synth_init_(net)
    HP.set_precision(net, prec)
    x = torch.empty((2, 1, 238, 608, 968), device=dev)
    mask = torch.empty((2, 1, 608, 968), device=dev)
    for i in range(2):
        engine.synth_fill_(x[i], 1234 + i)
        engine.synth_fill_(mask[i], 4321 + i, mode=1, thr=0.9)
    crit = torch.nn.BCEWithLogitsLoss()
    def step(model):
        for p in net.parameters():
            p.grad = None
        crit(model(x), mask).backward()

    times = {"plain": [], "stock_ddp": [], "plain_b": [], "stock_ddp_bucket_view": []}
    for plain_key, key, kw in (("plain", "stock_ddp", {}), ("plain_b", "stock_ddp_bucket_view", {"gradient_as_bucket_view": True})):
        ddp = DDP(net, device_ids=[0], broadcast_buffers=False, **kw)     # (one reducer at a time: its hooks sit on the parameters)
        for r in range(4):
            for k, model in ((plain_key, net), (key, ddp)):
                autograd.SEGMENT_AUTO = model is ddp
                for _ in range(3):
                    step(model)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    step(model)
                torch.cuda.synchronize()
                times[k].append((time.perf_counter() - t0) / 20 * 1e3)
        del ddp
        gc.collect()
        for p in net.parameters():
            p.grad = None
    autograd.SEGMENT_AUTO = True
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    out["modes"][prec] = {"ms_per_step": {k: [round(t, 3) for t in v] for k, v in times.items()}, "median_ms": {k: round(v, 3) for k, v in med.items()},
                          "cubes_per_s": {k: round(2e3 / v, 2) for k, v in med.items()}, "stock_ddp_over_plain": round(med["stock_ddp"] / med["plain"], 4),
                          "stock_ddp_bucket_view_over_plain": round(med["stock_ddp_bucket_view"] / med["plain_b"], 4),
                          "chain": list(autograd.LAST_PLAN)}
    del net, x, mask
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
dist.destroy_process_group()
