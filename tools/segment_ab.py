#!/usr/bin/env python3
"""What the chain of autograd nodes costs (autograd.run_staged): the benched step (CubeNET-64, two 238x608x968 cubes) with the
network as ONE node against the same tape cut into the chain a process group selects, arms interleaved in one process, no
collective anywhere.  Also checks that both arms give bit-identical gradients.
usage: segment_ab.py [rounds] [steps] > profiles/r05_segment_ab.json"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import hyperpri_amd as HP  # noqa: E402
from hyperpri_amd import autograd, engine  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
out = {"rounds": rounds, "steps": steps, "library_stamp": bench._lib_stamp(), "modes": {}}
for prec in ("fp32", "bf16"):
    net = HP.CubeNET(238, 1, first_depth=64, bilinear=False).to(dev).train()
    bench.synth_init_(net)
    HP.set_precision(net, prec)
    x = torch.empty((2, 1, 238, 608, 968), device=dev)
    mask = torch.empty((2, 1, 608, 968), device=dev)
    for i in range(2):
        engine.synth_fill_(x[i], 1234 + i)
        engine.synth_fill_(mask[i], 4321 + i, mode=1, thr=0.9)
    crit = torch.nn.BCEWithLogitsLoss()
    sd = {k: v.clone() for k, v in net.state_dict().items()}

    def step():
        for p in net.parameters():
            p.grad = None
        loss = crit(net(x), mask)
        loss.backward()
        return loss

    grads = {}
    for arm in (True, "segmented"):
        net.fused_tape = arm
        net.load_state_dict(sd)
        step()
        torch.cuda.synchronize()
        grads[arm] = [p.grad.clone() for p in net.parameters()]
    same = all(torch.equal(a, b) for a, b in zip(grads[True], grads["segmented"]))
    times = {"one_node": [], "chain": []}
    for r in range(rounds):
        for arm, key in ((True, "one_node"), ("segmented", "chain")):
            net.fused_tape = arm
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            times[key].append((time.perf_counter() - t0) / steps * 1e3)
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    out["modes"][prec] = {"ms_per_step": {k: [round(t, 3) for t in v] for k, v in times.items()}, "median_ms": {k: round(v, 3) for k, v in med.items()},
                          "chain_over_one_node": round(med["chain"] / med["one_node"], 4), "plan": list(autograd.LAST_PLAN),
                          "gradients_bit_identical": same}
    del net, x, mask, grads
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
