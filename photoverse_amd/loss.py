"""The ArcFace identity loss of the training step (``/root/reference/models/loss.py:9-78`` ``FaceLoss``,
``/root/reference/models/arcface_resnet.py:12-133`` ``ArcFaceResNet18``; used at ``train.py:339-341,521-535``) on the HIP kernels.

``FaceLoss(device, 'arcface')(x, x_gen, maximize=True, normalize=True)`` has the reference's call surface.  The network is frozen and in
eval mode, so every BatchNorm that FOLLOWS a convolution / the Linear is folded into its weights; the ones in FRONT of a zero-padded
convolution (``IRBlock.bn0``, ``bn4``) stay explicit per-channel affines.  The 64-channel stage runs zero-padded to the MFMA GEMM's
128-column tile.  1x1 stride-2 shortcut convolutions run as the 3x3 stride-2 kernel with only the centre tap set.  The gradient with
respect to the GENERATED image (what ``accelerator.backward`` needs, train.py:536) comes from the same tape mechanism as the rest of the
training step (``tape.py``): ``attach`` hangs the loss off an image buffer of a larger plan; ``loss_and_grad`` is the stand-alone form.

``model_name='facenet'`` ([EXT] facenet_pytorch InceptionResnetV1, loss.py:25) is not available.  ``pretrained`` weights are a network
download in the reference (arcface_resnet.py:130-140): load them with ``model.load_state_dict`` from a local file instead.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn

from .ops import require_cuda
from .tape import Tape, Var, conv3_dgrad_weight


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the arithmetic runs in FaceLoss' launch plan")


class IRBlock(_Holder):
    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.bn0 = nn.BatchNorm2d(inplanes)
        self.conv1 = nn.Conv2d(inplanes, inplanes, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(inplanes)
        self.prelu = nn.PReLU()
        self.conv2 = nn.Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class ArcFaceResNet18(_Holder):
    """Parameter layout of the reference's ``ArcFaceResNet18`` (``use_se=False``): its state dict loads by name."""

    def __init__(self, layers=(2, 2, 2, 2), image_size=128):
        super().__init__()
        self.image_size = image_size
        self.inplanes = 64
        self.conv1 = nn.Conv2d(1, 64, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.prelu = nn.PReLU()
        self.layer1 = self._make_layer(64, layers[0])
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.bn4 = nn.BatchNorm2d(512)
        self.fc5 = nn.Linear(512 * (image_size // 16) ** 2, 512)
        self.bn5 = nn.BatchNorm1d(512)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.xavier_normal_(m.weight)
                if getattr(m, "bias", None) is not None:
                    nn.init.constant_(m.bias, 0)
        for p in self.parameters():
            p.requires_grad_(False)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, 1, stride=stride, bias=False), nn.BatchNorm2d(planes))
        layers = [IRBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(IRBlock(planes, planes))
        return nn.Sequential(*layers)


# ---------------------------------------------------------------------------------------------------------------- weight preparation
def _bn_affine(bn, pad_to=None):
    s = (bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps))
    t = bn.bias.detach().float() - bn.running_mean.detach().float() * s
    if pad_to is not None and pad_to > s.numel():
        z = torch.zeros(pad_to - s.numel(), device=s.device)
        s, t = torch.cat([s, z]), torch.cat([t, z])
    return s.contiguous(), t.contiguous()


def _fold(conv_w, bn, cout_p, cin_p):
    """conv (no bias) followed by an eval-mode BatchNorm -> (weight [cout_p, cin_p, k, k] fp32 zero padded, bias [cout_p])."""
    s, t = _bn_affine(bn)
    w = conv_w.detach().float() * s.view(-1, 1, 1, 1)
    co, ci, kh, kw = w.shape
    wp = torch.zeros(cout_p, cin_p, kh, kw, device=w.device)
    wp[:co, :ci] = w
    bp = torch.zeros(cout_p, device=w.device)
    bp[:co] = t
    return wp, bp.contiguous()


def _conv3_rows(w):
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(torch.float16).contiguous()


def _pad(c):
    return max(128, c)


class FaceLoss(nn.Module):
    def __init__(self, device, model_name: str = "arcface", input_size: Optional[int] = None, model: Optional[ArcFaceResNet18] = None):
        super().__init__()
        if model_name != "arcface":
            raise NotImplementedError(f"FaceLoss model {model_name!r}: only 'arcface' is built ('facenet' is [EXT] facenet_pytorch)")
        self.device = torch.device(device)
        self.model_name = model_name
        self.input_size = input_size or 128                 # loss.py:15
        self.model = (model if model is not None else ArcFaceResNet18(image_size=self.input_size)).to(self.device).eval()
        self._plans: Dict[tuple, object] = {}

    # ------------------------------------------------------------------ the network on a tape
    def _block(self, tp: Tape, x: Var, blk: IRBlock, batch, h, w):
        cin, cout = blk.conv1.in_channels, blk.conv2.out_channels
        cip, cop = _pad(cin), _pad(cout)
        slope = blk.prelu.weight.detach().float().contiguous()
        s0, t0 = _bn_affine(blk.bn0, cip)
        x0 = tp.col_affine(x, s0, t0)
        w1, b1 = _fold(blk.conv1.weight, blk.bn1, cip, cip)
        y = tp.conv3(x0, _conv3_rows(w1), conv3_dgrad_weight(w1), bias=b1, batch=batch, h=h, w=w, colstats=False)
        y = tp.prelu(y, slope)
        w2, b2 = _fold(blk.conv2.weight, blk.bn2, cop, cip)
        res = x
        if blk.downsample is not None:
            wd_, bd = _fold(blk.downsample[0].weight, blk.downsample[1], cop, cip)          # 1x1 (stride) conv + BN ...
            w3 = torch.zeros(cop, cip, 3, 3, device=wd_.device)
            w3[:, :, 1, 1] = wd_[:, :, 0, 0]                                                   # ... as the centre tap of a 3x3
            res = tp.conv3(x, _conv3_rows(w3), conv3_dgrad_weight(w3), bias=bd, batch=batch, h=h, w=w, stride=blk.stride, colstats=False)
        y = tp.conv3(y, _conv3_rows(w2), conv3_dgrad_weight(w2), bias=b2, batch=batch, h=h, w=w, stride=blk.stride, residual=res, colstats=False)
        if blk.stride == 2:
            h, w = h // 2, w // 2
        return tp.prelu(y, slope), h, w

    def embed(self, tp: Tape, img: torch.Tensor, *, normalize: bool, want_grad: bool, unscale: float = 1.0, holder=None):
        """img: fp32 (B, 3, H, W) buffer (or (B, 1, H, W) gray).  Returns (embedding Var [B, 512] fp16, holder whose ``.g`` becomes the
        fp32 gradient w.r.t. ``img`` when the backward plan is built)."""
        m, S = self.model, self.input_size
        B, C, H, W = img.shape
        if C != 3:
            raise NotImplementedError("FaceLoss expects RGB images (B, 3, H, W)")
        mul, add = ((1.0 / 127.5, -1.0) if normalize else (1.0, 0.0))
        gray = tp.rf.gray_resize(img, size=S, mul=mul, add=add)
        cols = tp.rf.im2col3x3(gray, batch=B, cin=1, h=S, wd=S, kpad=64)
        w1, b1 = _fold(m.conv1.weight, m.bn1, 128, 1)                                          # [128, 1, 3, 3]
        w1c = torch.zeros(128, 64, dtype=torch.float16, device=img.device)
        w1c[:, :9] = w1.reshape(128, 9).to(torch.float16)
        holder = holder if holder is not None else SimpleNamespace(g=None)
        y = tp.linear(Var(cols, False), w1c, w1c.t().contiguous(), bias=b1, rows_per_image=S * S)
        y.needs = want_grad

        def conv1_bwd(y=y):
            if y.g is None or not want_grad:
                return
            # d/d(gray): a 128 -> 1 channel 3x3 convolution of dY with the flipped filter (pv_conv_out, two zero output channels)
            wdg = torch.zeros(3, 3, 3, 128, device=img.device)
            wdg[0] = w1.flip(2, 3).permute(1, 2, 3, 0)[0]
            assert y.g.is_contiguous()
            dg3 = tp.rb.conv_out(y.g, wdg.reshape(3, -1).to(torch.float16).contiguous(), None, batch=B, cin=128, h=S, wd=S, cout=3)
            holder.g = tp.rb.gray_resize_backward(dg3[:, :1], h=H, w=W, mul=mul * unscale)
        tp.back.append(conv1_bwd)
        x = tp.prelu(y, m.prelu.weight.detach().float().contiguous())
        x = tp.maxpool2x2(x, batch=B, h=S, w=S)
        h = w = S // 2
        for layer in (m.layer1, m.layer2, m.layer3, m.layer4):
            for blk in layer:
                x, h, w = self._block(tp, x, blk, B, h, w)
        s4, t4 = _bn_affine(m.bn4)
        x = tp.col_affine(x, s4, t4)                                                           # Dropout: eval mode (loss.py:16)
        flat = tp.reshape(x, (B, h * w * 512))
        s5, t5 = _bn_affine(m.bn5)
        wf = m.fc5.weight.detach().float().view(512, 512, h, w).permute(0, 2, 3, 1).reshape(512, -1) * s5.view(-1, 1)   # NCHW flatten -> NHWC rows
        bf = (m.fc5.bias.detach().float() * s5 + t5).contiguous()
        emb = tp.linear(flat, *tp.frozen(wf), bias=bf)
        return emb, holder

    def attach(self, tp: Tape, img_real: torch.Tensor, img_gen: torch.Tensor, *, weight: float = 1.0, maximize: bool = True, normalize: bool = True,
               unscale: float = 1.0, holder=None):
        """Hang the loss off ``img_gen`` (a buffer some other part of ``tp``'s forward plan writes).  Returns a namespace: ``loss`` (device
        scalar, fp32), ``per_sample``, ``dimg`` (holder: ``.g`` = unscale * weight * grad_scale * d loss / d img_gen after
        ``tp.build_backward()``; the fp16 gradients in between carry the tape's loss scale).  ``holder``: write that gradient into an existing
        holder instead (e.g. ``vae_train.decode_on_tape(...).dimg``, chaining the decoder's backward behind the loss)."""
        e1, _ = self.embed(tp, img_real, normalize=normalize, want_grad=False)
        e2, holder = self.embed(tp, img_gen, normalize=normalize, want_grad=True, unscale=unscale, holder=holder)
        per_sample, de2 = tp.rf.cosine_embedding_loss(e1.t, e2.t, target=1.0 if maximize else -1.0, gscale=weight * tp.S)
        loss = tp.rf.reduce_mean(per_sample, mode="mean")

        def seed():
            e2.g = de2
        tp.back.append(seed)
        return SimpleNamespace(loss=loss, per_sample=per_sample, dimg=holder, emb_real=e1, emb_gen=e2)

    # ------------------------------------------------------------------ stand-alone use
    def _plan(self, B, H, W, maximize, normalize):
        tp = Tape(self.device, 16384.0)        # image-level gradients are ~1e-7: fp16 storage in between needs the loss scale
        x = tp.rf.hold(torch.zeros((B, 3, H, W), dtype=torch.float32, device=self.device))
        xg = tp.rf.hold(torch.zeros((B, 3, H, W), dtype=torch.float32, device=self.device))
        out = self.attach(tp, x, xg, maximize=maximize, normalize=normalize, unscale=1.0 / tp.S)
        tp.build_backward()
        return SimpleNamespace(tape=tp, x=x, xg=xg, out=out)

    @torch.no_grad()
    def loss_and_grad(self, x, x_gen, maximize=True, normalize=True):
        """(loss, d loss / d x_gen) - fp32 device tensors."""
        require_cuda(x_gen, "x_gen")
        B, _, H, W = x_gen.shape
        key = (B, H, W, bool(maximize), bool(normalize))
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = self._plan(B, H, W, maximize, normalize)
        plan.x.copy_(x)
        plan.xg.copy_(x_gen)
        plan.tape.rf.run()
        plan.tape.rb.run()
        return plan.out.loss.clone(), plan.out.dimg.g.clone()

    def forward(self, x, x_gen, maximize=True, normalize=True):
        """loss.py:64-78: the cosine embedding loss between the ArcFace embeddings of ``x`` and ``x_gen`` (value only)."""
        return self.loss_and_grad(x, x_gen, maximize, normalize)[0]
