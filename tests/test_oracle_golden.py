"""The CPU oracle (oracle/hyperpri_oracle.py) against every fixture captured from the real
reference modules by tests/golden/make_golden.py.  CPU only."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import hyperpri_oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")
TOL = dict(rtol=1e-5, atol=1e-6)


def _load(name):
    return np.load(os.path.join(G, name + ".npz"))


def _u(seed, shape):
    return torch.from_numpy(O._u(seed, int(np.prod(shape))).reshape(shape).copy())


def _block(name, shapes, fwd, seed0, nin):
    """Replay a block fixture through the oracle: train fwd+bwd with the stored dOut, then eval."""
    z = _load(name)
    sd = O.synth_state_dict(shapes, seed0=seed0, bn_random=True)
    work = OrderedDict()
    leaves = OrderedDict()
    for k, v in sd.items():
        if O.is_param(k):
            leaves[k] = work[k] = v.clone().requires_grad_(True)
        else:
            work[k] = v.clone()
    xs = [torch.from_numpy(z[f"in{i}"]).requires_grad_(True) for i in range(nin)]
    out = fwd(work, *xs, True)
    np.testing.assert_allclose(out.detach().numpy(), z["out_train"], **TOL)
    (out * torch.from_numpy(z["dout"])).sum().backward()
    for i, x in enumerate(xs):
        np.testing.assert_allclose(x.grad.numpy(), z[f"din{i}"], rtol=1e-4, atol=1e-6)
    for k, t in leaves.items():
        np.testing.assert_allclose(t.grad.numpy(), z["grad/" + k], rtol=1e-4, atol=2e-6, err_msg=k)
    for k in z.files:
        if k.startswith("buf/"):
            np.testing.assert_allclose(work[k[4:]].numpy(), z[k], **TOL, err_msg=k)
    with torch.no_grad():
        oe = fwd(work, *[x.detach() for x in xs], False)
    np.testing.assert_allclose(oe.numpy(), z["out_eval"], **TOL)


def _dc_shapes(cin, cout, mid=None, p=""):
    sd = OrderedDict()
    O._dc_keys(sd, p, cin, cout, mid)
    return sd


def test_block_doubleconv():
    _block("block_doubleconv", _dc_shapes(5, 7), lambda sd, x, t: O.double_conv(sd, "", x, t), 2000, 1)


def test_block_doubleconv_mid():
    _block("block_doubleconv_mid", _dc_shapes(6, 4, 9), lambda sd, x, t: O.double_conv(sd, "", x, t), 2100, 1)


def test_block_down():
    _block("block_down", _dc_shapes(4, 6, p="maxpool_conv.1."), lambda sd, x, t: O.down(sd, "", x, t), 2200, 1)


@pytest.mark.parametrize("name,cin,cout,seed", [("block_up", 8, 4, 2300), ("block_up_big", 64, 32, 2400)])
def test_block_up(name, cin, cout, seed):
    sd = OrderedDict()
    O._up_keys(sd, "", cin, cout)
    _block(name, sd, lambda s, x1, x2, t: O.up(s, "", x1, x2, t), seed, 2)


@pytest.mark.parametrize("name,bil,att,seed", [("block_up_bilinear", True, False, 2600), ("block_up_attn", False, True, 2700),
                                               ("block_up_bilinear_attn", True, True, 2800)])
def test_block_up_variants(name, bil, att, seed):
    sd = OrderedDict()
    O._up_keys(sd, "", 16, 8, bil, att)
    _block(name, sd, lambda s, x1, x2, t: O.up(s, "", x1, x2, t, bil, att), seed, 2)


@pytest.mark.parametrize("name,bil,seed", [("block_up_crop", False, 3100), ("block_up_crop_mixed", False, 3200),
                                           ("block_up_bilinear_crop", True, 3300)])
def test_block_up_crop(name, bil, seed):
    """Skip smaller than the upsampled tensor: F.pad crops (model_parts.py:73-80); fixtures from make_golden_crop.py."""
    sd = OrderedDict()
    O._up_keys(sd, "", 16, 8, bil, False)
    _block(name, sd, lambda s, x1, x2, t: O.up(s, "", x1, x2, t, bil, False), seed, 2)


def test_block_outconv():
    sd = OrderedDict()
    O._conv_keys(sd, "conv", 2, 6, 1)
    _block("block_outconv", sd, lambda s, x, t: O.out_conv(s, "", x), 2500, 1)


NETS = [
    ("net_unet3_tiny", lambda: O.unet_shapes(3, 1), O.unet_forward, {}, 1234, (2, 3, 36, 50), 4321),
    ("net_cubenet64_tiny", lambda: O.cubenet_shapes(6, 1, 64), O.cubenet_forward, {"first_depth": 64}, 1235, (2, 1, 6, 36, 50), 4321),
    ("net_cubenet128_tiny", lambda: O.cubenet_shapes(6, 1, 128), O.cubenet_forward, {"first_depth": 128}, 1236, (2, 1, 6, 36, 50), 4321),
    ("net_unet3_bilinear_tiny", lambda: O.unet_shapes(3, 1, True), O.unet_forward, {"bilinear": True}, 1239, (2, 3, 36, 50), 4321),
    ("net_unet3_attn_tiny", lambda: O.unet_shapes(3, 1, True, True), O.unet_forward, {"bilinear": True, "use_attention": True}, 1240, (2, 3, 36, 50), 4321),
    ("net_cubenet64_bilinear_tiny", lambda: O.cubenet_shapes(6, 1, 64, True), O.cubenet_forward, {"first_depth": 64, "bilinear": True}, 1241, (2, 1, 6, 36, 50), 4321),
    ("net_spectral_tiny", lambda: O.spectral_shapes(10, 1, 4), O.spectral_forward, {}, 1237, (3, 10, 7, 9), 4322),
    ("net_spectral_f48", lambda: O.spectral_shapes(22, 1, 48), O.spectral_forward, {}, 1238, (2, 22, 12, 20), 4323),
    ("net_spectral_3class", lambda: O.spectral_shapes(10, 3, 4), O.spectral_forward, {}, 1252, (2, 10, 7, 9), 4332),
    ("net_spectral_f50", lambda: O.spectral_shapes(22, 1, 50), O.spectral_forward, {}, 1242, (2, 22, 9, 14), 4324),
    # SpectralUNET(bnorm=False) (models.py:72,105-110): Linear -> ReLU stages, no BatchNorm entries in the state dict
    ("net_spectral_nobn_tiny", lambda: O.spectral_shapes(10, 1, 4, bnorm=False), O.spectral_forward, {}, 1237, (3, 10, 7, 9), 4322),
    ("net_spectral_nobn_f50", lambda: O.spectral_shapes(22, 1, 50, bnorm=False), O.spectral_forward, {}, 1242, (2, 22, 9, 14), 4324),
    # the BASELINE configs' exact channel widths at reduced spatial size (tests/golden/make_golden_widths.py)
    ("net_spectral1650_small", lambda: O.spectral_shapes(238, 1, 1650), O.spectral_forward, {}, 1250, (2, 238, 16, 24), 4330),
    ("net_cubenet128_300_small", lambda: O.cubenet_shapes(300, 1, 128), O.cubenet_forward, dict(first_depth=128), 1251,
     (2, 1, 300, 32, 48), 4331),
]
MASK_THR = {"net_spectral1650_small": 0.8, "net_cubenet128_300_small": 0.9}


@pytest.mark.parametrize("name,shapes,fwd,kw,xseed,xshape,mseed", NETS, ids=[n[0] for n in NETS])
def test_tiny_net(name, shapes, fwd, kw, xseed, xshape, mseed):
    z = _load(name)
    sd = O.synth_state_dict(shapes())
    x = _u(xseed, xshape)
    thr = MASK_THR.get(name, 0.9 if "spectral" not in name else 0.7)
    mask = (_u(mseed, (xshape[0], int(z["logits"].shape[1])) + tuple(xshape[-2:])) > thr).float()
    logits, loss, grads = O.train_step(fwd, sd, x, mask, **kw)
    np.testing.assert_allclose(logits.numpy(), z["logits"], rtol=1e-4, atol=2e-6)
    assert abs(loss - float(z["loss"])) < 1e-6
    acc, dice, iou = O.seg_metrics(logits, mask)
    assert round(dice, 4) == round(float(z["dice"]), 4) and round(iou, 4) == round(float(z["iou"]), 4)
    names = list(z["grad_names"])
    assert names == list(grads.keys())
    for i, k in enumerate(names):
        g = grads[k].double().flatten()
        assert abs(float(g.norm()) - z["grad_l2"][i]) <= 1e-4 * z["grad_l2"][i] + 1e-9, k
        n = min(16, g.numel())
        np.testing.assert_allclose(g[:n].float().numpy(), z["grad_head"][i][:n], rtol=2e-3, atol=1e-7, err_msg=k)
    for k in z.files:
        if k.startswith("buf/"):
            np.testing.assert_allclose(sd[k[4:]].numpy(), z[k], rtol=1e-5, atol=1e-6, err_msg=k)
    # eval-mode forward with the updated running stats
    with torch.no_grad():
        le = fwd(sd, x, train=False, **kw)
    np.testing.assert_allclose(le.numpy(), z["logits_eval"], rtol=1e-4, atol=2e-6)


def test_cubenet_conv2d_equivalent():
    """first layer as Conv2d over D channels == the reference's Conv3d."""
    sd = O.synth_state_dict(O.cubenet_shapes(6, 1, 64))
    sd2 = OrderedDict((k, v.clone()) for k, v in sd.items())
    x = _u(1235, (2, 1, 6, 36, 50))
    with torch.no_grad():
        a = O.cubenet_forward(sd, x, 64, True, conv3d=True)
        b = O.cubenet_forward(sd2, x, 64, True, conv3d=False)
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=1e-4, atol=1e-5)


def test_known_answers():
    with open(os.path.join(G, "known_answers.json")) as f:
        known = json.load(f)
    c = known["counts"]
    # README.md:65 "~31.0M parameters"; test_models.ipynb:201 "38 param tensors / 30388051 elements"
    assert c["UNet(3,1)"]["elements"] == 31043521 and c["UNet(3,1)"]["tensors"] == 82
    assert c["SpectralUNET(238,1,1650)"] == {"tensors": 38, "elements": 30388051, "state_dict_keys": 65}
    assert c["CubeNET(238,1,64)"]["elements"] == 31178881 and c["CubeNET(238,1,64)"]["state_dict_keys"] == 138
    assert c["CubeNET(300,1,128)"]["elements"] == 31608961
    for nm, shp in [("unet3_full", O.unet_shapes(3, 1)), ("cubenet64_full", O.cubenet_shapes(238, 1, 64)),
                    ("cubenet128_full", O.cubenet_shapes(300, 1, 128)), ("spectral1650_full", O.spectral_shapes(238, 1, 1650))]:
        assert list(shp.keys()) == known[nm]["keys"], nm
        assert [list(s) for s in shp.values()] == known[nm]["shapes"], nm
        n = sum(int(np.prod(s)) for k, s in shp.items() if O.is_param(k))
        assert n == {"unet3_full": 31043521, "cubenet64_full": 31178881, "cubenet128_full": 31608961,
                     "spectral1650_full": 30388051}[nm]


def test_generator_known_answer():
    # canonical splitmix64: first output for state 0 is 0xE220A8397B1DCDAF -> top 24 bits / 2^24
    v = O._u(0, 2)
    assert v[0] == 0.0  # mix(0) == 0
    assert O._u(1, 1)[0] == np.float32((0xE220A8397B1DCDAF >> 40) / 2.0 ** 24)


def test_oracle_gradients_vs_fp64_reference_samples():
    """The deep gradient fixtures (tests/golden/make_golden_grads.py): the oracle's fp32 gradients of the tiny CubeNET-64, sampled at
    the fixture's positions, equal the reference modules' fp32 samples to rounding, and both sit within the recorded distance of the
    fp64 samples -- so the GPU test that measures the HIP path against fp64 is anchored on the reference, not on the oracle."""
    import numpy as np
    z = np.load(os.path.join(G, "grads_cubenet64_tiny.npz"))
    sd = O.synth_state_dict(O.cubenet_shapes(6, 1, 64))
    x = torch.from_numpy(O._u(1235, 2 * 6 * 36 * 50).reshape(2, 1, 6, 36, 50).copy())
    m = (torch.from_numpy(O._u(4321, 2 * 36 * 50).reshape(2, 1, 36, 50).copy()) > 0.9).float()
    _, loss, grads = O.train_step(O.cubenet_forward, sd, x, m, first_depth=64)
    assert abs(loss - float(z["loss32"])) < 1e-6
    names = list(z["grad_names"])
    ns = int(z["ns"])
    for k, nm in enumerate(names):
        g = grads[nm if nm in grads else nm.replace("inc.0.", "first_conv.")].reshape(-1)
        numel = g.numel()
        idx = np.arange(numel) if numel <= ns else np.minimum((O._u(9000 + k, ns).astype(np.float64) * numel).astype(np.int64), numel - 1)
        cnt = int(z["grad_sample_count"][k])
        got = g[torch.from_numpy(idx)].double().numpy()
        ref32 = z["grad_sample32"][k, :cnt].astype(np.float64)
        scale = max(float(np.abs(ref32).max()), 1e-6)
        assert float(np.abs(got - ref32).max()) <= 2e-3 * scale + 1e-7, nm
