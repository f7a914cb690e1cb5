"""Deep gradient fixtures from the REAL reference modules (build container only; never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_grads.py [tiny] [c2b1] [c2b2] [c5]

The network fixtures of make_golden.py pin every gradient tensor by its L2 norm and its first 16 values.  These add, per
parameter tensor, the gradient at NS pseudo-random positions (counter-based generator, so the positions are not stored),
computed twice by the reference modules: in fp32 (what the reference itself produces) and in fp64 (``module.double()``,
the arithmetic both fp32 results approximate).  A test can then attribute a difference to the reference's own fp32
summation noise or to the kernels: ||g_hip - g64|| against ||g_ref32 - g64||, per tensor, in relative L2 over the sample.

  grads_<net>_tiny        the four tiny networks of make_golden.py (same inputs / weights), NS = 1024 per tensor
  grads_cubenet64_full_b1 CubeNET(238,1,64) @ (1,1,238,608,968)  -- the existing full-size fixture's step, NS = 256
  grads_cubenet64_full_b2 CubeNET(238,1,64) @ (2,1,238,608,968)  -- the BENCHED shape (BatchNorm over two cubes), with logits
                          sub-sample / loss / Dice / IoU / BN buffers like the other full-size fixtures, NS = 256
  grads_cubenet128_300_full_b1  config C5's network, NS = 256 (fp64 pass needs ~40 GB: run alone)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as MG  # noqa: E402  (sets sys.path for the oracle and the reference)

O = MG.O


def sample_index(k, numel, ns):
    """Positions of tensor number k (registration order): floor(u(9000+k, i) * numel), i < ns; all positions when numel <= ns."""
    if numel <= ns:
        return np.arange(numel, dtype=np.int64)
    return np.minimum((O._u(9000 + k, ns).astype(np.float64) * numel).astype(np.int64), numel - 1)


def run(mod, x, mask, dtype):
    mod = mod.to(dtype)
    mod.train()
    for p in mod.parameters():
        p.grad = None
    logits = mod(x.to(dtype))
    loss = torch.nn.BCEWithLogitsLoss()(logits, mask.to(dtype))
    loss.backward()
    return logits.detach(), float(loss.detach())


def grads_fixture(name, make_mod, x, mask, ns, extras=False, stride=97):
    rec = {}
    samples = {}
    for dtype, tag in ((torch.float32, "32"), (torch.float64, "64")):
        mod = make_mod()
        MG.load_synth(mod, 1000, False)
        logits, loss = run(mod, x, mask, dtype)
        rec["loss" + tag] = np.float64(loss)
        names = []
        for k, (nm, p) in enumerate(mod.named_parameters()):
            g = p.grad.detach().reshape(-1)
            idx = torch.from_numpy(sample_index(k, g.numel(), ns))
            samples.setdefault(nm, {})[tag] = g[idx].double().numpy()
            samples[nm]["l2_" + tag] = float(g.double().norm())
            names.append(nm)
        if extras and tag == "32":
            acc, dice, iou = O.seg_metrics(logits, mask)
            lg = logits.numpy()
            rec.update({"mean": np.float64(logits.double().mean().item()), "std": np.float64(logits.double().std().item()),
                        "acc": np.float64(acc), "dice": np.float64(dice), "iou": np.float64(iou),
                        "logits_sub": lg.reshape(-1)[::stride].copy(), "stride": np.int64(stride)})
            for kb, b in MG.buffers_of(mod).items():
                if b.size <= 4096:
                    rec["buf/" + kb] = b
        del mod, logits
    rec["grad_names"] = np.array(names)
    rec["ns"] = np.int64(ns)
    rec["grad_l2_32"] = np.array([samples[n]["l2_32"] for n in names], dtype=np.float64)
    rec["grad_l2_64"] = np.array([samples[n]["l2_64"] for n in names], dtype=np.float64)
    width = max(len(samples[n]["32"]) for n in names)
    g32 = np.zeros((len(names), width), dtype=np.float32)
    g64 = np.zeros((len(names), width), dtype=np.float64)
    cnt = np.zeros(len(names), dtype=np.int64)
    for i, n in enumerate(names):
        m = len(samples[n]["32"])
        g32[i, :m], g64[i, :m], cnt[i] = samples[n]["32"], samples[n]["64"], m
    rec["grad_sample32"], rec["grad_sample64"], rec["grad_sample_count"] = g32, g64, cnt
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    noise = [float(np.linalg.norm(g32[i, :cnt[i]] - g64[i, :cnt[i]]) / (np.linalg.norm(g64[i, :cnt[i]]) + 1e-300)) for i in range(len(names))]
    print("wrote", name, "loss32", rec["loss32"], "loss64", rec["loss64"], "max rel fp32-vs-fp64 over tensors with a gradient",
          max(v for v, l in zip(noise, rec["grad_l2_64"]) if l > 1e-6), flush=True)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["tiny"]
    RM = MG.RM
    if "tiny" in which:
        h, w = 36, 50
        m = (MG.u(4321, (2, 1, h, w)) > 0.9).float()
        grads_fixture("grads_unet3_tiny", lambda: RM.UNet(3, 1, bilinear=False), MG.u(1234, (2, 3, h, w)), m, 1024)
        grads_fixture("grads_cubenet64_tiny", lambda: RM.CubeNET(6, 1, first_depth=64, bilinear=False), MG.u(1235, (2, 1, 6, h, w)), m, 1024)
        grads_fixture("grads_cubenet128_tiny", lambda: RM.CubeNET(6, 1, first_depth=128, bilinear=False), MG.u(1236, (2, 1, 6, h, w)), m, 1024)
        m3 = (MG.u(4322, (3, 1, 7, 9)) > 0.7).float()
        grads_fixture("grads_spectral_tiny", lambda: RM.SpectralUNET(10, 1, 4), MG.u(1237, (3, 10, 7, 9)), m3, 1024)
    H, W = 608, 968
    if "c2b1" in which:
        mk = (MG.u(4321, (1, 1, H, W)) > 0.9).float()
        grads_fixture("grads_cubenet64_full_b1", lambda: RM.CubeNET(238, 1, first_depth=64, bilinear=False),
                      MG.u(1234, (1, 1, 238, H, W)), mk, 256)
    if "c2b2" in which:
        x = torch.cat([MG.u(1234 + n, (1, 1, 238, H, W)) for n in range(2)], 0)        # bench.py's cubes: seeds 1234 + n
        mk = torch.cat([(MG.u(4321 + n, (1, 1, H, W)) > 0.9).float() for n in range(2)], 0)
        grads_fixture("grads_cubenet64_full_b2", lambda: RM.CubeNET(238, 1, first_depth=64, bilinear=False), x, mk, 256, extras=True)
    if "c5" in which:
        mk = (MG.u(4321, (1, 1, H, W)) > 0.9).float()
        grads_fixture("grads_cubenet128_300_full_b1", lambda: RM.CubeNET(300, 1, first_depth=128, bilinear=False),
                      MG.u(1234, (1, 1, 300, H, W)), mk, 256)


if __name__ == "__main__":
    main()
