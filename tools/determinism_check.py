#!/usr/bin/env python3
"""Bit-reproducibility of a full-size step with steps in flight: the same forward + backward N times without a fence in between,
gradients of every step compared bit for bit with the first one's (a buffer recycled while the weight-gradient stream still read it
would show here).  usage: determinism_check.py [precision] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import hyperpri_amd as HP
from hyperpri_amd import engine

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
net = HP.CubeNET(238, 1, first_depth=64, bilinear=False).to(dev).train()
bench.synth_init_(net)
HP.set_precision(net, prec)
x = torch.empty((2, 1, 238, 608, 968), device=dev)
mask = torch.empty((2, 1, 608, 968), device=dev)
for i in range(2):
    engine.synth_fill_(x[i], 1234 + i)
    engine.synth_fill_(mask[i], 4321 + i, mode=1, thr=0.9)
sd = {k: v.clone() for k, v in net.state_dict().items()}
grads = []
for s in range(steps):
    net.load_state_dict(sd)                      # (BatchNorm running statistics back to the start: same step every time)
    for p in net.parameters():
        p.grad = None
    _, loss = HP.forward_loss(net, x, mask)
    loss.backward()
    grads.append([p.grad for p in net.parameters()])      # kept alive, compared at the end: no fence inside the loop
torch.cuda.synchronize()
bad = 0
for s in range(1, steps):
    for (n, _), a, b in zip(net.named_parameters(), grads[0], grads[s]):
        if not torch.equal(a, b):
            bad += 1
            print("step", s, n, "differs: max", float((a - b).abs().max()))
print(prec, steps, "steps:", "bit-identical" if bad == 0 else f"{bad} tensors differ", "loss", float(loss.detach()))
sys.exit(1 if bad else 0)
