"""SD-v1.5 ``AutoencoderKL`` on HIP kernels: DECODE - the step right after the denoising loop
(``/root/reference/models/infer.py:121-123``: ``vae.decode(latents / vae.config.scaling_factor).sample.clamp(-1, 1)``;
the model is loaded at ``/root/reference/models/modeling_utils.py:56``), SURVEY.md section 8f row 1 - and ENCODE
(``vae.encode(pixel_values).latent_dist.sample()``, ``infer.py:63`` for ``from_noised_image``, ``train.py:471``).

Parameter names follow diffusers (``post_quant_conv``, ``decoder.conv_in``, ``decoder.mid_block.resnets.{0,1}``,
``decoder.mid_block.attentions.0.{group_norm,to_q,to_k,to_v,to_out.0}``, ``decoder.up_blocks.i.resnets.j``,
``decoder.up_blocks.i.upsamplers.0.conv``, ``decoder.conv_norm_out``, ``decoder.conv_out``; ``quant_conv``, ``encoder.conv_in``,
``encoder.down_blocks.i.{resnets.j,downsamplers.0.conv}``, ``encoder.mid_block``, ``encoder.conv_norm_out``, ``encoder.conv_out``)
so the HF ``vae`` checkpoint loads.  Everything runs on the UNet's kernels:
3x3 convs = implicit-GEMM ``pv_gemm_conv`` (nearest-x2 upsample folded into the gather), GroupNorm(+SiLU) kernels, and the
mid block's single-head attention over the H*W tokens (head dim 512, too wide for the flash kernel's register tiles) as
GEMM (Q.K^T per image) -> ``pv_softmax_rows`` -> GEMM (P.V with V^T produced directly by an operand-swapped GEMM; the V bias
is added after P.V, exact because softmax rows sum to 1).  Images are decoded in sub-batches so that no operand exceeds the
2 GiB the buffer descriptors address.  Encoder specifics: ``conv_in`` (3 channels) is im2col + GEMM like the decoder's; the
Downsample2D convs (``F.pad(x,(0,1,0,1))`` + stride 2 / padding 0) are ``pv_gemm_conv`` with ``pad=0``; ``quant_conv`` (1x1, linear)
is folded into ``conv_out``'s weights in fp32 when the plan is built, and the 8 moment channels come out of one fp32-output GEMM
whose weight rows are zero-padded to the 128-column tile.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict

import torch
import torch.nn as nn

from .ops import ACT_NONE, ACT_SILU, Recorder, require_cuda

SD15_VAE_CONFIG = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                       norm_num_groups=32, scaling_factor=0.18215)


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise NotImplementedError(f"{type(self).__name__} only holds parameters")


class _Res(_Holder):
    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None


class _Attn(_Holder):
    def __init__(self, ch, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(ch, ch), nn.Linear(ch, ch), nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Dropout(0.0)])


class _Mid(_Holder):
    def __init__(self, ch, groups):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(ch, ch, groups), _Res(ch, ch, groups)])
        self.attentions = nn.ModuleList([_Attn(ch, groups)])


class _Upsample(_Holder):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)


class _Up(_Holder):
    def __init__(self, cin, cout, layers, groups, add_up):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.upsamplers = nn.ModuleList([_Upsample(cout)]) if add_up else None


class _Downsample(_Holder):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=0)


class _Down(_Holder):
    def __init__(self, cin, cout, layers, groups, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([_Downsample(cout)]) if add_down else None


class _Encoder(_Holder):
    def __init__(self, in_channels, latent_channels, boc, layers_per_block, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        out = boc[0]
        for i in range(len(boc)):
            prev, out = out, boc[i]
            self.down_blocks.append(_Down(prev, out, layers_per_block, groups, i != len(boc) - 1))
        self.mid_block = _Mid(boc[-1], groups)
        self.conv_norm_out = nn.GroupNorm(groups, boc[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[-1], 2 * latent_channels, 3, padding=1)


class DiagonalGaussianDistribution:
    """Mirror of diffusers' posterior object: ``mean``, ``logvar`` (clamped to [-30, 20]), ``std``, ``sample()``, ``mode()``."""

    def __init__(self, moments: torch.Tensor):
        self.parameters = moments
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None):
        eps = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)   # the draw (input), not arithmetic
        if self.parameters.is_cuda and self.parameters.dtype == torch.float32:
            from .ops import Recorder
            rec = Recorder(self.parameters.device)
            out = rec.posterior_sample(self.parameters.contiguous(), eps.contiguous())
            rec.run()
            return out
        return self.mean + self.std * eps

    def mode(self):
        return self.mean


class _Decoder(_Holder):
    def __init__(self, latent_channels, out_channels, boc, layers_per_block, groups):
        super().__init__()
        self.conv_in = nn.Conv2d(latent_channels, boc[-1], 3, padding=1)
        self.mid_block = _Mid(boc[-1], groups)
        rev = list(reversed(boc))
        self.up_blocks = nn.ModuleList()
        out = rev[0]
        for i in range(len(boc)):
            prev, out = out, rev[i]
            self.up_blocks.append(_Up(prev, out, layers_per_block + 1, groups, i != len(boc) - 1))
        self.conv_norm_out = nn.GroupNorm(groups, boc[0], eps=1e-6)
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)


def _f16(t):
    return t.detach().to(torch.float16).contiguous()


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _conv3_w(w):
    return w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(torch.float16).contiguous()


class AutoencoderKL(nn.Module):
    """The SD VAE.  ``decode(z) -> .sample`` (B,3,8h,8w) fp32 like diffusers' ``DecoderOutput``; ``encode(x) -> .latent_dist``
    (``with_encoder=False`` builds the decoder half only)."""

    MAX_OPERAND_BYTES = 1 << 30     # keep every activation operand well under the 2 GiB of a buffer descriptor

    def __init__(self, **overrides):
        super().__init__()
        cfg = dict(SD15_VAE_CONFIG)
        cfg.update(overrides)
        cfg.setdefault("with_encoder", True)
        cfg.setdefault("in_channels", 3)
        self.config = SimpleNamespace(**cfg)
        self.post_quant_conv = nn.Conv2d(cfg["latent_channels"], cfg["latent_channels"], 1)
        self.decoder = _Decoder(cfg["latent_channels"], cfg["out_channels"], tuple(cfg["block_out_channels"]), cfg["layers_per_block"],
                                cfg["norm_num_groups"])
        if cfg["with_encoder"]:
            self.quant_conv = nn.Conv2d(2 * cfg["latent_channels"], 2 * cfg["latent_channels"], 1)
            self.encoder = _Encoder(cfg["in_channels"], cfg["latent_channels"], tuple(cfg["block_out_channels"]),
                                    cfg["layers_per_block"], cfg["norm_num_groups"])
        self._plans: Dict[tuple, object] = {}

    def repack(self):
        self._plans.clear()

    def load_state_dict(self, sd, strict=True, **k):
        # a decoder-only model ignores the encoder / quant_conv keys of an HF vae checkpoint; a full model accepts a
        # decoder-only state dict (its encoder keeps its current weights) but never a partial encoder
        enc = ("encoder.", "quant_conv.")
        if not self.config.with_encoder:
            sd = {k_: v for k_, v in sd.items() if not k_.startswith(enc)}
        elif not any(k_.startswith(enc) for k_ in sd):
            sd = dict(sd)
            sd.update({k_: v for k_, v in self.state_dict().items() if k_.startswith(enc)})
        r = super().load_state_dict(sd, strict=strict, **k)
        self.repack()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._plans = {}
        return r

    # ------------------------------------------------------------------ plan
    def _sub_batch(self, batch, h, w):
        boc = self.config.block_out_channels
        up = 2 ** (len(boc) - 1)
        worst = max(boc[0], boc[1] if len(boc) > 1 else boc[0]) * (h * up) * (w * up) * 2      # bytes per image of the largest tensor
        return max(1, min(batch, self.MAX_OPERAND_BYTES // worst))

    def _res(self, rec: Recorder, m: _Res, x, b, h, w):
        geo = dict(batch=b, hin=h, win=w, hout=h, wout=w)
        g = self.config.norm_num_groups
        hn = rec.groupnorm(x, _f32(m.norm1.weight), _f32(m.norm1.bias), batch=b, hw=h * w, eps=m.norm1.eps, act=ACT_SILU, groups=g)
        h1 = rec.gemm(hn, _conv3_w(m.conv1.weight), bias=_f32(m.conv1.bias), conv=geo, colstats=True)
        h2 = rec.groupnorm(h1, _f32(m.norm2.weight), _f32(m.norm2.bias), batch=b, hw=h * w, eps=m.norm2.eps, act=ACT_SILU, groups=g)
        sc = x
        if m.conv_shortcut is not None:
            sc = rec.gemm(x, _f16(m.conv_shortcut.weight.reshape(m.conv_shortcut.out_channels, -1)), bias=_f32(m.conv_shortcut.bias),
                          rows_per_image=h * w)
        return rec.gemm(h2, _conv3_w(m.conv2.weight), bias=_f32(m.conv2.bias), residual=sc, conv=geo, colstats=True)

    def _attn(self, rec: Recorder, m: _Attn, x, b, h, w):
        n, c = h * w, m.to_q.in_features
        g = rec.groupnorm(x, _f32(m.group_norm.weight), _f32(m.group_norm.bias), batch=b, hw=n, eps=m.group_norm.eps, act=ACT_NONE,
                          groups=self.config.norm_num_groups)
        q = rec.gemm(g, _f16(m.to_q.weight), bias=_f32(m.to_q.bias), rows_per_image=n)    # [b*n, c]
        k = rec.gemm(g, _f16(m.to_k.weight), bias=_f32(m.to_k.bias), rows_per_image=n)    # contiguous rows: used as a [N][K] weight
        wv = _f16(m.to_v.weight)
        o = rec.empty((b * n, c))
        scores = rec.empty((n, n))                                            # reused by every image (launches are stream ordered)
        vt = rec.empty((c, n))
        for i in range(b):
            rows = slice(i * n, (i + 1) * n)
            rec.gemm(q[rows], rec.hold(k[rows]), out=scores, splitk=0)        # S = Q K^T
            rec.softmax_rows(scores, scale=c ** -0.5)
            rec.gemm(wv, rec.hold(g[rows]), out=vt, splitk=0)                  # V^T = Wv X^T    (no bias: added after P.V)
            rec.gemm(scores, vt, bias=_f32(m.to_v.bias), out=o[rows], splitk=0)   # O = P V + bv
        return rec.gemm(o, _f16(m.to_out[0].weight), bias=_f32(m.to_out[0].bias), residual=x, rows_per_image=n, colstats=True)

    def _plan(self, sb, h, w, dev):
        cfg, d = self.config, self.decoder
        rec = Recorder(dev)
        lc = cfg.latent_channels
        z = rec.empty((sb, lc, h, w), torch.float32)
        zq = rec.pointwise_nchw(z, _f32(self.post_quant_conv.weight.reshape(lc, lc)), _f32(self.post_quant_conv.bias), batch=sb, cin=lc,
                                cout=lc, hw=h * w)
        c_in = d.conv_in.out_channels
        kin, kpad = lc * 9, (lc * 9 + 63) // 64 * 64
        cols = rec.im2col3x3(zq, batch=sb, cin=lc, h=h, wd=w, kpad=kpad)
        w_in = torch.zeros(c_in, kpad, dtype=torch.float16, device=dev)
        w_in[:, :kin] = d.conv_in.weight.detach().reshape(c_in, kin).to(torch.float16)
        x = rec.gemm(cols, w_in, bias=_f32(d.conv_in.bias), rows_per_image=h * w, colstats=True)
        x = self._res(rec, d.mid_block.resnets[0], x, sb, h, w)
        x = self._attn(rec, d.mid_block.attentions[0], x, sb, h, w)
        x = self._res(rec, d.mid_block.resnets[1], x, sb, h, w)
        for blk in d.up_blocks:
            for r in blk.resnets:
                x = self._res(rec, r, x, sb, h, w)
            if blk.upsamplers is not None:
                conv = blk.upsamplers[0].conv
                x = rec.gemm(x, _conv3_w(conv.weight), bias=_f32(conv.bias),
                             conv=dict(batch=sb, hin=h, win=w, hout=2 * h, wout=2 * w, upsample=1), colstats=True)
                h, w = 2 * h, 2 * w
        xn = rec.groupnorm(x, _f32(d.conv_norm_out.weight), _f32(d.conv_norm_out.bias), batch=sb, hw=h * w, eps=d.conv_norm_out.eps,
                           act=ACT_SILU, groups=cfg.norm_num_groups)
        co = d.conv_out.out_channels
        wo = d.conv_out.weight.detach().permute(0, 2, 3, 1).reshape(co, -1).to(torch.float16).contiguous()
        img = rec.conv_out(xn, wo, _f32(d.conv_out.bias), batch=sb, cin=d.conv_out.in_channels, h=h, wd=w, cout=co)
        return SimpleNamespace(rec=rec, z=z, img=img)

    def _plan_encode(self, sb, H, W, dev):
        cfg, e = self.config, self.encoder
        rec = Recorder(dev)
        cin = e.conv_in.in_channels
        x_in = rec.empty((sb, cin, H, W), torch.float32)
        c0 = e.conv_in.out_channels
        kin, kpad = cin * 9, (cin * 9 + 63) // 64 * 64
        cols = rec.im2col3x3(x_in, batch=sb, cin=cin, h=H, wd=W, kpad=kpad)
        w_in = torch.zeros(c0, kpad, dtype=torch.float16, device=dev)
        w_in[:, :kin] = e.conv_in.weight.detach().reshape(c0, kin).to(torch.float16)
        x = rec.gemm(cols, w_in, bias=_f32(e.conv_in.bias), rows_per_image=H * W, colstats=True)
        h, w = H, W
        for blk in e.down_blocks:
            for r in blk.resnets:
                x = self._res(rec, r, x, sb, h, w)
            if blk.downsamplers is not None:
                conv = blk.downsamplers[0].conv
                x = rec.gemm(x, _conv3_w(conv.weight), bias=_f32(conv.bias),
                             conv=dict(batch=sb, hin=h, win=w, hout=h // 2, wout=w // 2, stride=2, pad=0), colstats=True)
                h, w = h // 2, w // 2
        x = self._res(rec, e.mid_block.resnets[0], x, sb, h, w)
        x = self._attn(rec, e.mid_block.attentions[0], x, sb, h, w)
        x = self._res(rec, e.mid_block.resnets[1], x, sb, h, w)
        xn = rec.groupnorm(x, _f32(e.conv_norm_out.weight), _f32(e.conv_norm_out.bias), batch=sb, hw=h * w, eps=e.conv_norm_out.eps,
                           act=ACT_SILU, groups=cfg.norm_num_groups)
        # quant_conv (1x1) o conv_out (3x3) is one linear map: fold it in fp32, pad the 2*latent output rows to one 128-column tile
        nm = e.conv_out.out_channels
        wq = self.quant_conv.weight.detach().reshape(nm, nm).to(torch.float32)
        wc = e.conv_out.weight.detach().permute(0, 2, 3, 1).reshape(nm, -1).to(torch.float32)
        w_fold = torch.zeros(128, wc.shape[1], dtype=torch.float16, device=dev)
        w_fold[:nm] = (wq @ wc).to(torch.float16)
        b_fold = torch.zeros(128, dtype=torch.float32, device=dev)
        b_fold[:nm] = wq @ e.conv_out.bias.detach().to(torch.float32) + self.quant_conv.bias.detach().to(torch.float32)
        mom = rec.gemm(xn, w_fold, bias=b_fold, out_f32=True, conv=dict(batch=sb, hin=h, win=w, hout=h, wout=w))   # [sb*h*w][128] fp32
        return SimpleNamespace(rec=rec, x=x_in, moments=mom, h=h, w=w, nm=nm)

    def _sub_batch_encode(self, batch, H, W):
        boc = self.config.block_out_channels
        worst = max(boc[0] * H * W * 2, 64 * H * W * 2)      # first-level activations / the im2col rows of conv_in
        return max(1, min(batch, self.MAX_OPERAND_BYTES // worst))

    def encode(self, x: torch.Tensor):
        """``x``: (B,3,H,W) pixels in [-1,1] on the GPU -> ``.latent_dist`` over (B,latent_channels,H/8,W/8), fp32."""
        if not self.config.with_encoder:
            raise RuntimeError("this AutoencoderKL was built with with_encoder=False")
        require_cuda(x, "pixel_values")
        B, _, H, W = x.shape
        down = 2 ** (len(self.config.block_out_channels) - 1)
        assert H % down == 0 and W % down == 0, (H, W, down)
        sb = self._sub_batch_encode(B, H, W)
        key = ("enc", sb, H, W, x.device)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = self._plan_encode(sb, H, W, x.device)
        outs = []
        for i in range(0, B, sb):
            chunk = x[i:i + sb].to(torch.float32)
            n = chunk.shape[0]
            plan.x[:n].copy_(chunk)
            plan.rec.run()
            m = plan.moments.view(sb, plan.h, plan.w, -1)[:n, :, :, :plan.nm]
            outs.append(m.permute(0, 3, 1, 2).contiguous())
        return SimpleNamespace(latent_dist=DiagonalGaussianDistribution(torch.cat(outs, 0)))

    def decode(self, z: torch.Tensor):
        require_cuda(z, "latents")
        B, _, h, w = z.shape
        sb = self._sub_batch(B, h, w)
        key = (sb, h, w, z.device)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = self._plan(sb, h, w, z.device)
        outs = []
        for i in range(0, B, sb):
            chunk = z[i:i + sb].to(torch.float32)
            n = chunk.shape[0]
            plan.z[:n].copy_(chunk)
            plan.rec.run()
            outs.append(plan.img[:n].clone())
        return SimpleNamespace(sample=torch.cat(outs, 0).to(z.dtype))
