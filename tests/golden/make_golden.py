"""Generate the golden fixtures under tests/golden/ from the REAL reference modules.

Run only in the build container (needs /root/reference, read-only; never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--full]

For every fixture the reference module (``src/Experiments/model_parts.py`` / ``models.py``) is
constructed, generator-defined weights (oracle.synth_state_dict) are loaded with
``load_state_dict``, the module is run on generator-defined inputs and its outputs, gradients and
BatchNorm buffers are written to ``*.npz``.  Only data (inputs' seeds, expected outputs) is stored;
no reference source text is copied.
"""
import argparse
import json
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from oracle import hyperpri_oracle as O  # noqa: E402
from src.Experiments import model_parts as RP  # noqa: E402  (reference)
from src.Experiments import models as RM  # noqa: E402  (reference)


def shapes_of(mod):
    return OrderedDict((k, tuple(v.shape)) for k, v in mod.state_dict().items())


def load_synth(mod, seed0, bn_random):
    shp = shapes_of(mod)
    sd = O.synth_state_dict(shp, seed0=seed0, bn_random=bn_random)
    mod.load_state_dict(sd)
    return shp


def u(seed, shape):
    return torch.from_numpy(O._u(seed, int(np.prod(shape))).reshape(shape).copy())


def grads_of(mod):
    return OrderedDict((k, p.grad.detach().numpy().copy()) for k, p in mod.named_parameters())


def buffers_of(mod):
    return OrderedDict((k, b.detach().numpy().copy()) for k, b in mod.named_buffers())


def block_fixture(name, mod, inputs, seed0):
    """Train-mode fwd + bwd with dOut = (u - 0.5), then an eval-mode fwd with the updated stats."""
    load_synth(mod, seed0, bn_random=True)
    mod.train()
    xs = [x.clone().requires_grad_(True) for x in inputs]
    out = mod(*xs)
    r = u(seed0 + 500, out.shape) - 0.5
    (out * r).sum().backward()
    rec = {"out_train": out.detach().numpy(), "dout": r.numpy()}
    for i, x in enumerate(xs):
        rec[f"in{i}"] = inputs[i].numpy()
        rec[f"din{i}"] = x.grad.numpy()
    for k, g in grads_of(mod).items():
        rec["grad/" + k] = g
    for k, b in buffers_of(mod).items():
        rec["buf/" + k] = b
    mod.eval()
    with torch.no_grad():
        rec["out_eval"] = mod(*inputs).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name, {k: v.shape for k, v in rec.items() if k.startswith("out")})


def summarize_grads(mod):
    out = OrderedDict()
    for k, p in mod.named_parameters():
        g = p.grad.detach().double().flatten()
        out[k] = {"l2": float(g.norm()), "head": [float(v) for v in g[:16].float()]}
    return out


def net_fixture(name, mod, x, mask, seed0=1000, full_logits=True, stride=None):
    shp = load_synth(mod, seed0, bn_random=False)
    mod.train()
    logits = mod(x)
    loss = torch.nn.BCEWithLogitsLoss()(logits, mask)
    loss.backward()
    acc, dice, iou = O.seg_metrics(logits, mask)
    rec = {"loss": np.float64(loss.item()), "mean": np.float64(logits.double().mean().item()),
           "std": np.float64(logits.double().std().item()),
           "acc": np.float64(acc), "dice": np.float64(dice), "iou": np.float64(iou)}
    lg = logits.detach().numpy()
    if full_logits:
        rec["logits"] = lg
    else:
        rec["logits_sub"] = lg.reshape(-1)[::stride].copy()
        rec["stride"] = np.int64(stride)
    for k, b in buffers_of(mod).items():
        if b.size <= 4096:
            rec["buf/" + k] = b
    gs = summarize_grads(mod)
    rec["grad_names"] = np.array(list(gs.keys()))
    rec["grad_l2"] = np.array([v["l2"] for v in gs.values()], dtype=np.float64)
    rec["grad_head"] = np.array([v["head"] + [0.0] * (16 - len(v["head"])) for v in gs.values()], dtype=np.float32)
    mod.eval()
    with torch.no_grad():
        le = mod(x).numpy()
    if full_logits:
        rec["logits_eval"] = le
    else:
        rec["logits_eval_sub"] = le.reshape(-1)[::stride].copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print("wrote", name, "loss", loss.item(), "dice", dice, "iou", iou)
    return shp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also the 608x968 full-size fixtures (minutes, ~8 GB)")
    args = ap.parse_args()
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # (i) per-block fixtures --------------------------------------------------------------
    block_fixture("block_doubleconv", RP.DoubleConv(5, 7), [u(11, (2, 5, 9, 11))], 2000)
    block_fixture("block_doubleconv_mid", RP.DoubleConv(6, 4, 9), [u(12, (1, 6, 7, 35))], 2100)
    block_fixture("block_down", RP.Down(4, 6), [u(13, (2, 4, 9, 11))], 2200)
    block_fixture("block_up", RP.Up(8, 4, bilinear=False), [u(14, (1, 8, 3, 5)), u(15, (1, 4, 6, 11))], 2300)
    block_fixture("block_up_big", RP.Up(64, 32, bilinear=False), [u(16, (2, 64, 5, 37)), u(17, (2, 32, 11, 75))], 2400)
    block_fixture("block_outconv", RP.OutConv(6, 2), [u(18, (2, 6, 5, 7))], 2500)
    # Up variants no HyperPRI experiment configures (SURVEY.md 8 a3'): bilinear upsampling and the "attention" product
    block_fixture("block_up_bilinear", RP.Up(16, 8, bilinear=True), [u(19, (2, 8, 5, 9)), u(20, (2, 8, 11, 19))], 2600)
    block_fixture("block_up_attn", RP.Up(16, 8, bilinear=False, use_attention=True),
                  [u(21, (2, 16, 5, 9)), u(22, (2, 8, 11, 19))], 2700)
    block_fixture("block_up_bilinear_attn", RP.Up(16, 8, bilinear=True, use_attention=True),
                  [u(23, (2, 8, 5, 9)), u(24, (2, 8, 10, 18))], 2800)

    # (ii) tiny full networks -------------------------------------------------------------
    known = OrderedDict()
    h, w = 36, 50
    m = (u(4321, (2, 1, h, w)) > 0.9).float()
    shp = net_fixture("net_unet3_tiny", RM.UNet(3, 1, bilinear=False), u(1234, (2, 3, h, w)), m)
    known["unet3"] = {"keys": list(shp.keys()), "shapes": [list(s) for s in shp.values()]}
    shp = net_fixture("net_cubenet64_tiny", RM.CubeNET(6, 1, first_depth=64, bilinear=False),
                      u(1235, (2, 1, 6, h, w)), m)
    known["cubenet64_d6"] = {"keys": list(shp.keys()), "shapes": [list(s) for s in shp.values()]}
    shp = net_fixture("net_cubenet128_tiny", RM.CubeNET(6, 1, first_depth=128, bilinear=False),
                      u(1236, (2, 1, 6, h, w)), m)
    known["cubenet128_d6"] = {"keys": list(shp.keys()), "shapes": [list(s) for s in shp.values()]}
    net_fixture("net_unet3_bilinear_tiny", RM.UNet(3, 1, bilinear=True), u(1239, (2, 3, h, w)), m)
    net_fixture("net_unet3_attn_tiny", RM.UNet(3, 1, bilinear=True, use_attention=True), u(1240, (2, 3, h, w)), m)
    net_fixture("net_cubenet64_bilinear_tiny", RM.CubeNET(6, 1, first_depth=64, bilinear=True),
                u(1241, (2, 1, 6, h, w)), m)
    # (CubeNET(first_depth=128, bilinear=True) cannot run in the reference: models.py:196 builds DoubleConv(256,..)
    #  for a 192-channel concat)
    m3 = (u(4322, (3, 1, 7, 9)) > 0.7).float()
    shp = net_fixture("net_spectral_tiny", RM.SpectralUNET(10, 1, 4), u(1237, (3, 10, 7, 9)), m3)
    known["spectral_10_4"] = {"keys": list(shp.keys()), "shapes": [list(s) for s in shp.values()]}
    shp = net_fixture("net_spectral_f48", RM.SpectralUNET(22, 1, 48), u(1238, (2, 22, 12, 20)),
                      (u(4323, (2, 1, 12, 20)) > 0.7).float())

    # F = 50 (not a multiple of 4, like the configured 1650): exercises the unaligned concat offset
    net_fixture("net_spectral_f50", RM.SpectralUNET(22, 1, 50), u(1242, (2, 22, 9, 14)),
                (u(4324, (2, 1, 9, 14)) > 0.7).float())

    # (iv) known answers --------------------------------------------------------------------
    def count(mod):
        ps = list(mod.parameters())
        return {"tensors": len(ps), "elements": int(sum(p.numel() for p in ps)),
                "state_dict_keys": len(mod.state_dict())}
    known["counts"] = {
        "UNet(3,1)": count(RM.UNet(3, 1, bilinear=False)),
        "CubeNET(238,1,64)": count(RM.CubeNET(238, 1, 64, bilinear=False)),
        "CubeNET(300,1,128)": count(RM.CubeNET(300, 1, 128, bilinear=False)),
        "SpectralUNET(238,1,1650)": count(RM.SpectralUNET(238, 1, 1650)),
    }
    for nm, mod in [("unet3", RM.UNet(3, 1, bilinear=False)),
                    ("cubenet64", RM.CubeNET(238, 1, 64, bilinear=False)),
                    ("cubenet128", RM.CubeNET(300, 1, 128, bilinear=False)),
                    ("spectral1650", RM.SpectralUNET(238, 1, 1650))]:
        shp = shapes_of(mod)
        known[nm + "_full"] = {"keys": list(shp.keys()), "shapes": [list(s) for s in shp.values()]}
    # default-init pin: torch.manual_seed(7) -> first/last few values of a few tensors
    torch.manual_seed(7)
    mod = RM.UNet(3, 1, bilinear=False)
    sd = mod.state_dict()
    known["init_seed7_unet3"] = {k: [float(v) for v in sd[k].flatten()[:4]]
                                 for k in ["inc.double_conv.0.weight", "up1.up.weight", "outc.conv.bias"]}
    torch.manual_seed(7)
    mod = RM.CubeNET(6, 1, 64, bilinear=False)
    sd = mod.state_dict()
    known["init_seed7_cubenet64_d6"] = {k: [float(v) for v in sd[k].flatten()[:4]]
                                        for k in ["first_conv.weight", "inc2.0.weight", "up4.up.bias", "outc.conv.weight"]}
    torch.manual_seed(7)
    mod = RM.SpectralUNET(10, 1, 4)
    sd = mod.state_dict()
    known["init_seed7_spectral_10_4"] = {k: [float(v) for v in sd[k].flatten()[:4]]
                                         for k in ["tail.0.weight", "up2.0.weight", "outc.bias"]}
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(known, f, indent=0)
    print("wrote known_answers.json")

    # (iii) full-size -------------------------------------------------------------------------
    if args.full:
        H, W = 608, 968
        mk = (u(4321, (1, 1, H, W)) > 0.9).float()
        x = u(1234, (1, 1, 238, H, W))
        net_fixture("net_cubenet64_full", RM.CubeNET(238, 1, first_depth=64, bilinear=False), x, mk,
                    full_logits=False, stride=97)
        del x
        mk2 = torch.cat([mk, (u(4322, (1, 1, H, W)) > 0.9).float()], 0)
        x = torch.cat([u(1234, (1, 3, H, W)), u(1235, (1, 3, H, W))], 0)
        net_fixture("net_unet3_full", RM.UNet(3, 1, bilinear=False), x, mk2, full_logits=False, stride=97)


if __name__ == "__main__":
    main()
