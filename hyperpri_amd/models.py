"""Drop-in replacements for the reference's networks (src/Experiments/models.py): ``UNet`` :23-68,
``SpectralUNET`` :71-145, ``CubeNET`` :148-247, ``initialize_model`` :250-276,
``translate_load_dir`` :279-292.  Constructor signatures, attribute names, parameter registration
order and ``state_dict`` keys (including CubeNET's aliased ``first_conv.*`` / ``inc.0.*``) match the
reference, so ``params_HyperPRI.py``, ``PLTrainer.py`` and saved checkpoints work unchanged; the
forward/backward arithmetic runs in hand-written HIP kernels (see ``engine``).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import engine as E
from .autograd import run, run_staged
from .model_parts import *  # noqa: F401,F403  (the reference's callers star-import everything)
from .model_parts import DoubleConv, Down, OutConv, Up, _drain, has_hooks, skip_room


def _stage_params(net):
    """Per stage of ``net._stages()`` the parameters its modules own, each parameter once (CubeNET's ``first_conv`` is also
    ``inc[0]``), and all of ``net.parameters()`` accounted for."""
    seen, out = set(), []
    for mods in net._stages():
        ps = []
        for m in mods:
            for p in m.parameters():
                if id(p) not in seen:
                    seen.add(id(p))
                    ps.append(p)
        out.append(ps)
    if any(id(p) not in seen for p in net.parameters()):
        raise RuntimeError("hyperpri_amd: internal error: a parameter of the network belongs to no stage of its tape program")
    return out


def set_parameter_requires_grad(model, feature_extraction):
    if feature_extraction:
        for p in model.parameters():
            p.requires_grad = False


class UNet(nn.Module):
    def __init__(self, n_channels, n_classes, bilinear=True, feature_extraction=False, use_attention=False,
                 analyze=False):
        super(UNet, self).__init__()
        self.n_channels, self.n_classes = n_channels, n_classes
        self.bilinear, self.use_attention, self.analyze = bilinear, use_attention, analyze
        factor = 2 if bilinear else 1
        w = [64 * 2 ** i for i in range(5)]          # 64,128,256,512,1024 (models.py:34-39)
        self.inc = DoubleConv(n_channels, w[0])
        self.down1 = Down(w[0], w[1])
        self.down2 = Down(w[1], w[2])
        self.down3 = Down(w[2], w[3])
        self.down4 = Down(w[3], w[4] // factor)
        self.up1 = Up(w[4], w[3], bilinear, use_attention=use_attention)
        self.up2 = Up(w[3], w[2], bilinear, use_attention=use_attention)
        self.up3 = Up(w[2], w[1], bilinear, use_attention=use_attention)
        self.up4 = Up(w[1], w[0] * factor, bilinear, use_attention=use_attention)
        self.outc = OutConv(w[0], n_classes)

    # One autograd node for the whole network (default): the skip tensors x1..x4 feed two consumers each, and inside
    # one tape their two gradient contributions are summed by the HIP kernels (accumulating epilogues) instead of by
    # autograd's ATen add.  Under a process group without a GradSync sink (stock DistributedDataParallel: Lightning
    # strategy="ddp", PLTrainer.py:434-442) the same tape is cut into a chain of a few nodes so that the reducer receives the
    # decoder's and the bottleneck's gradients while the encoder's backward still runs (autograd.run_staged; "segmented"
    # forces the chain).  ``fused_tape = False`` (or forward hooks on a child) calls the children one by one, each its
    # own node -- what a foreign composition of these modules gets anyway.
    fused_tape = True

    def _stages(self):
        """The modules of each stage of the tape program below, in order (their parameters leave together under a chain)."""
        st = []
        for m in (self.inc, self.down1, self.down2, self.down3, self.down4, self.up1, self.up2, self.up3, self.up4):
            st += m._stages()
        st[-1] = st[-1] + [self.outc]
        return st

    def forward(self, x):
        E.throttle(x.device)
        if self.fused_tape and not has_hooks(self):
            def prog(tape, a, need):
                # the four skip tensors are produced inside the buffers their concats will use (model_parts.py:87)
                x1 = yield from self.inc._gen(tape, a[0], need[0], room=skip_room(self.up4))
                yield
                x2 = yield from self.down1._gen(tape, x1, room=skip_room(self.up3))
                yield
                x3 = yield from self.down2._gen(tape, x2, room=skip_room(self.up2))
                yield
                x4 = yield from self.down3._gen(tape, x3, room=skip_room(self.up1))
                yield
                # (bf16 mode: what feeds a transposed convolution is also written as planes by its producer)
                cp = E.convt_planes_mode(self) and not self.bilinear and not self.use_attention
                x5 = yield from self.down4._gen(tape, x4, out_planes=cp)
                yield
                y = yield from self.up1._gen(tape, x5, x4, out_planes=cp)
                yield
                y = yield from self.up2._gen(tape, y, x3, out_planes=cp)
                yield
                y = yield from self.up3._gen(tape, y, x2, out_planes=cp)
                yield
                y = yield from self.up4._gen(tape, y, x1, head_next=True)          # (bf16 mode: the head reads its input as bf16 planes)
                return self.outc._ops(tape, y)
            logits = run_staged(prog, [x], _stage_params(self), self.fused_tape, input_planes=E.input_planes_for(self), name="unet",
                                lib_kind=getattr(self, "hpri_h16", None))
        else:
            x1 = self.inc(x)
            x2 = self.down1(x1)
            x3 = self.down2(x2)
            x4 = self.down3(x3)
            x5 = self.down4(x4)
            y = self.up1(x5, x4)
            y = self.up2(y, x3)
            y = self.up3(y, x2)
            y = self.up4(y, x1)
            logits = self.outc(y)
        if self.analyze:
            return (logits, logits, torch.sigmoid(logits))
        return logits


class SpectralUNET(torch.nn.Module):
    """Per-pixel MLP "U-Net" (models.py:71-145).  Every Linear is a 1x1 conv over the H*W pixels of an
    image; BatchNorm1d statistics are per image because the reference loops over images (:132)."""

    def __init__(self, hsi_depth, n_classes, bn_feats=16, bnorm=True):
        super(SpectralUNET, self).__init__()
        self.hsi_depth = self.n_channels = hsi_depth
        self.n_classes = n_classes
        self.layer_feats = [bn_feats] * 5
        self._bnorm = bnorm
        f = bn_feats
        self.tail = self._basic_module(hsi_depth, f, bn=bnorm)
        self.down1 = self._basic_module(f, f, bn=bnorm)
        self.down2 = self._basic_module(f, f, bn=bnorm)
        self.down3 = self._basic_module(f, f, bn=bnorm)
        self.down4 = self._basic_module(f, f, bn=bnorm)
        self.up1 = self._basic_module(f, f, bn=bnorm)
        self.up2 = self._basic_module(2 * f, f, bn=bnorm)
        self.up3 = self._basic_module(2 * f, f, bn=bnorm)
        self.up4 = self._basic_module(2 * f, f, bn=bnorm)
        self.outc = torch.nn.Linear(2 * f, self.n_classes)

    def _basic_module(self, in_feats, out_feats, bn=True):
        if not bn:
            return torch.nn.Sequential(torch.nn.Linear(in_feats, out_feats), torch.nn.ReLU())
        return torch.nn.Sequential(torch.nn.Linear(in_feats, out_feats), torch.nn.BatchNorm1d(out_feats),
                                   torch.nn.ReLU())

    def _layer(self, tape, x, seq, need_dx=True, **kw):
        bn = E.BNRef(seq[1]) if self._bnorm else None
        return E.conv_bn_relu(tape, x, seq[0].weight, seq[0].bias, bn, self.training, 1, groups=x.N, need_dx=need_dx,
                              precision=getattr(self, "hpri_precision", None), relu_without_bn=not self._bnorm, **kw)

    fused_tape = True       # see UNet.fused_tape (no per-child route here: the reference's forward is one loop over images)

    def _stages(self):
        return [[self.tail], [self.down1], [self.down2], [self.down3], [self.down4], [self.up1], [self.up2], [self.up3],
                [self.up4, self.outc]]

    def forward(self, x):
        E.throttle(x.device)
        def prog(tape, a, need):
            L = self._layer
            if E.plane_gemm_mode(self, self._bnorm) and (tape.record or self.training or not E.FOLD_EVAL_BN
                                                          or E.predict_gemm_planes_ok(getattr(self, "hpri_precision", None))):
                # bf16 mode: the three inner skips are concatenated on bf16 planes -- the producers of both halves write into one
                # padded plane buffer ([skip | zeros to a multiple of 32 | up]), the consumer's weight packs carry the gap -- and the
                # tensors between the layers exist as planes only (torch.cat of models.py:139-143 without a byte moved)
                f = self.layer_feats[0]
                hp = E.HEAD_PLANES
                x0 = L(tape, a[0], self.tail, need[0], **({"cat_room": f, "planes_only": True} if hp else {}))
                yield
                x1 = L(tape, x0, self.down1, cat_room=f, planes_only=True)
                yield
                x2 = L(tape, x1, self.down2, cat_room=f, planes_only=True)
                yield
                x3 = L(tape, x2, self.down3, cat_room=f, planes_only=True)
                yield
                x4 = L(tape, x3, self.down4, planes_only=True)
                yield
                t = L(tape, x4, self.up1, cat_into=x3, planes_only=True)
                yield
                c, gap = E.concat_planes(tape, x3, t)
                t = L(tape, c, self.up2, k_gap=gap, cat_into=x2, planes_only=True)
                yield
                c, gap = E.concat_planes(tape, x2, t)
                t = L(tape, c, self.up3, k_gap=gap, cat_into=x1, planes_only=True)
                yield
                c, gap = E.concat_planes(tape, x1, t)
                if hp:
                    # ... and so is the last one: the head reads the padded plane concat [tail | up4] through a weight row with the gap
                    t = L(tape, c, self.up4, k_gap=gap, cat_into=x0, planes_only=True)
                    c, gap = E.concat_planes(tape, x0, t)
                    return E.out_conv(tape, c, self.outc.weight, self.outc.bias, fuse_loss=self.n_classes == 1, k_gap=gap)
                t = L(tape, c, self.up4, k_gap=gap)
                return E.out_conv(tape, E.concat_channels(tape, x0, t), self.outc.weight, self.outc.bias, fuse_loss=self.n_classes == 1)
            x0 = L(tape, a[0], self.tail, need[0])
            yield
            x1 = L(tape, x0, self.down1)
            yield
            x2 = L(tape, x1, self.down2)
            yield
            x3 = L(tape, x2, self.down3)
            yield
            x4 = L(tape, x3, self.down4)
            yield
            t = L(tape, x4, self.up1)
            yield
            t = L(tape, E.concat_channels(tape, x3, t), self.up2)
            yield
            t = L(tape, E.concat_channels(tape, x2, t), self.up3)
            yield
            t = L(tape, E.concat_channels(tape, x1, t), self.up4)
            return E.out_conv(tape, E.concat_channels(tape, x0, t), self.outc.weight, self.outc.bias, fuse_loss=self.n_classes == 1)
        out = run_staged(prog, [x], _stage_params(self), self.fused_tape, name="spectral_unet", lib_kind=getattr(self, "hpri_h16", None))
        if self.n_classes != 1:
            # models.py:144 stores each image's (R*C, n_classes) result with .reshape(n_classes, R, C): the FLAT order is
            # pixel-major, class-minor.  `out` holds true class planes (N, K, R, C); re-order to the reference's element order
            # (a differentiable view + copy of N*K*R*C floats; no BASELINE config has n_classes != 1).
            n, k, r, c = out.shape
            out = out.reshape(n, k, r * c).transpose(1, 2).contiguous().reshape(n, k, r, c)
        return out


class CubeNET(torch.nn.Module):
    """UNet whose first layer is Conv3d(1 -> first_depth, (D,3,3)) over the whole spectrum
    (models.py:148-247) -- computed as a 3x3 conv over D input channels."""

    def __init__(self, hsi_depth, n_classes, first_depth=64, bilinear=True, use_attention=False, analyze=False):
        super(CubeNET, self).__init__()
        self.n_channels = 1
        self.depth, self.first_depth, self.n_classes = hsi_depth, first_depth, n_classes
        self.bilinear, self.use_attention, self.analyze = bilinear, use_attention, analyze
        factor = 2 if bilinear else 1
        self.first_conv = torch.nn.Conv3d(1, first_depth, kernel_size=(self.depth, 3, 3), padding=(0, 1, 1))
        self.inc = torch.nn.Sequential(self.first_conv, torch.nn.BatchNorm3d(first_depth), torch.nn.ReLU(inplace=True))
        self.inc2 = torch.nn.Sequential(torch.nn.Conv2d(first_depth, first_depth, kernel_size=3, padding=1),
                                        torch.nn.BatchNorm2d(first_depth), torch.nn.ReLU(inplace=True))
        C = 128
        self.down1 = Down(first_depth, C)
        self.down2 = Down(C, C * 2)
        self.down3 = Down(C * 2, C * 4)
        self.down4 = Down(C * 4, C * 8 // factor)
        self.up1 = Up(C * 8, C * 4, bilinear, use_attention=use_attention)
        self.up2 = Up(C * 4, C * 2, bilinear, use_attention=use_attention)
        self.up3 = Up(C * 2, C, bilinear, use_attention=use_attention)
        if first_depth == 64:
            self.up4 = Up(C, 64 * factor, bilinear, use_attention=use_attention)
        elif bilinear:
            self.upsample4 = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
            self.upconv4 = DoubleConv(C + first_depth, 64, 64)
        else:
            self.upsample4 = nn.ConvTranspose2d(C, 64, kernel_size=2, stride=2)
            self.upconv4 = DoubleConv(64 + first_depth, 64)
        self.outc = OutConv(64, self.n_classes)

    def _stem_gen(self, tape, x, need_dx, room=0):
        prec = getattr(self, "hpri_precision", None)
        h = E.conv_bn_relu(tape, x, self.first_conv.weight, self.first_conv.bias, E.BNRef(self.inc[1]),
                           self.training, 3, need_dx=need_dx, precision=prec, next_cout=self.inc2[0].weight.shape[0])
        yield
        return E.conv_bn_relu(tape, h, self.inc2[0].weight, self.inc2[0].bias, E.BNRef(self.inc2[1]),
                              self.training, 3, precision=prec, room=room)

    def _stem_ops(self, *args, **kw):
        return _drain(self._stem_gen(*args, **kw))

    def _stem(self, x):
        params = list(self.inc.parameters()) + list(self.inc2.parameters())
        return run(lambda tape, a, need: self._stem_ops(tape, a[0], need[0]), [x], params, name="cubenet_stem", lib_kind=getattr(self, "hpri_h16", None))

    def _up4_gen(self, tape, y, x1, need_dx1=True, head_next=False):
        """Last decoder stage: ``up4`` (first_depth 64) or the inline upsample4 -> pad -> cat -> upconv4 (models.py:229-240)."""
        if self.first_depth == 64:
            return self.up4._gen(tape, y, x1, need_dx1, head_next=head_next)
        w4 = None if self.bilinear else self.upsample4.weight
        b4 = None if self.bilinear else self.upsample4.bias
        cat = E.up_concat(tape, y, x1, w4, b4, need_dx1=need_dx1, precision=getattr(self, "hpri_precision", None))
        return self.upconv4._gen(tape, cat, head_next=head_next)

    def _up4_ops(self, *args, **kw):
        return _drain(self._up4_gen(*args, **kw))

    fused_tape = True       # see UNet.fused_tape

    def _stages(self):
        st = [[self.inc], [self.inc2]]
        for m in (self.down1, self.down2, self.down3, self.down4, self.up1, self.up2, self.up3):
            st += m._stages()
        if self.first_depth == 64:
            st += self.up4._stages()
        else:
            first, second = self.upconv4._stages()
            st += [[self.upsample4] + first, second]
        st[-1] = st[-1] + [self.outc]
        return st

    def forward(self, x):
        if x.dim() != 5 or x.shape[2] != self.depth:
            raise ValueError(f"CubeNET expects (N,1,{self.depth},R,C), got {tuple(x.shape)}")
        E.throttle(x.device)
        if self.fused_tape and not has_hooks(self):
            def prog(tape, a, need):
                up4 = self.up4 if self.first_depth == 64 else self.upsample4
                x1 = yield from self._stem_gen(tape, a[0], need[0], room=skip_room(up4))
                yield
                x2 = yield from self.down1._gen(tape, x1, room=skip_room(self.up3))
                yield
                x3 = yield from self.down2._gen(tape, x2, room=skip_room(self.up2))
                yield
                x4 = yield from self.down3._gen(tape, x3, room=skip_room(self.up1))
                yield
                cp = E.convt_planes_mode(self) and not self.bilinear and not self.use_attention     # see UNet.forward
                x5 = yield from self.down4._gen(tape, x4, out_planes=cp)
                yield
                y = yield from self.up1._gen(tape, x5, x4, out_planes=cp)
                yield
                y = yield from self.up2._gen(tape, y, x3, out_planes=cp)
                yield
                y = yield from self.up3._gen(tape, y, x2, out_planes=cp)
                yield
                y = yield from self._up4_gen(tape, y, x1, head_next=True)
                return self.outc._ops(tape, y)
            logits = run_staged(prog, [x], _stage_params(self), self.fused_tape, input_planes=E.input_planes_for(self, raw_ok=True), name="cubenet",
                                lib_kind=getattr(self, "hpri_h16", None))
        else:
            x1 = self._stem(x)
            x2 = self.down1(x1)
            x3 = self.down2(x2)
            x4 = self.down3(x3)
            x5 = self.down4(x4)
            y = self.up1(x5, x4)
            y = self.up2(y, x3)
            y = self.up3(y, x2)
            if self.first_depth == 64:
                y = self.up4(y, x1)
            else:
                params4 = list(self.upsample4.parameters()) + list(self.upconv4.parameters())
                y = run(lambda tape, a, need: self._up4_ops(tape, a[0], a[1], need[0]), [y, x1], params4, name="cubenet_up4",
                        lib_kind=getattr(self, "hpri_h16", None))
            logits = self.outc(y)
        if self.analyze:
            return (logits, logits, torch.sigmoid(logits))
        return logits


def initialize_model(model_name, num_classes, Network_parameters, analyze=False):
    """Name -> model factory with the reference's dictionary keys (models.py:250-276)."""
    p = Network_parameters
    if model_name == 'UNET':
        return UNet(p['channels'], num_classes, bilinear=p['bilinear'], feature_extraction=p['feature_extraction'],
                    use_attention=p['use_attention'], analyze=analyze)
    if model_name == 'SpectralUNET':
        return SpectralUNET(p['hsi_hi'] - p['hsi_lo'], num_classes, bn_feats=p['spectral_bn_size'])
    if model_name == 'CubeNET':
        return CubeNET(p['hsi_hi'] - p['hsi_lo'], num_classes, first_depth=p['3d_featmaps'], bilinear=p['bilinear'],
                       use_attention=p['use_attention'], analyze=analyze)
    raise RuntimeError('Invalid model')


def translate_load_dir(model_name, net_params):
    """Model name -> checkpoint directory name (models.py:279-292)."""
    if model_name == 'SpectralUNET':
        return f"{model_name}_{net_params['spectral_bn_size']}"
    if model_name == 'CubeNET':
        return f"{model_name}_{net_params['3d_featmaps']}"
    return "UNET"
