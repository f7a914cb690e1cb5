#!/usr/bin/env python3
"""Soak: N training steps (forward, loss, backward, fused Adam) of a mid-size CubeNET; prints memory every 10 steps and
checks that allocated memory is flat and the loss finite and decreasing.  usage: soak.py [steps] [precision]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hyperpri_amd as HP  # noqa: E402
from hyperpri_amd import engine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = HP.CubeNET(30, 1, first_depth=64, bilinear=False).to(dev).train()
HP.set_precision(net, prec)
x = torch.rand(2, 1, 30, 152, 242, device=dev)
mask = (torch.rand(2, 1, 152, 242, device=dev) > 0.9).float()
model = HP.SegmentationModel(net, optimizer="Adam", lr=1e-3)
opt = model.configure_optimizers()
mem, losses = [], []
for s in range(steps):
    opt.zero_grad()
    loss = model.training_step({"image": x, "mask": mask}, s)
    loss.backward()
    opt.step()
    if s % 10 == 9:
        torch.cuda.synchronize()
        m = model.epoch_metrics("tr")
        mem.append(torch.cuda.memory_allocated() >> 20)
        losses.append(m["tr_loss"])
        print(f"step {s + 1}: loss {m['tr_loss']:.5f} dice {m['tr_dice']:.4f} allocated {mem[-1]} MiB reserved {torch.cuda.memory_reserved() >> 20} MiB",
              flush=True)
assert all(torch.isfinite(torch.tensor(losses))), losses
assert losses[-1] < losses[0], losses
# no leak: the samples may wobble by a few MiB (cached packs and workspaces come and go with the allocator), but there must be
# no upward trend between the first and the second half of the run
half = len(mem) // 2
assert max(mem[1:]) - min(mem[1:]) <= 16 and abs(sum(mem[half:]) / len(mem[half:]) - sum(mem[1:half]) / max(len(mem[1:half]), 1)) <= 2, mem
print("soak ok")
