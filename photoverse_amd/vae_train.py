"""The VAE decoder as a differentiable piece of a training plan (``tape.py``): forward launches as ``vae.AutoencoderKL.decode``
plus the data gradient d(image)/d(latents).

Needed by the identity-loss branch of the training step (``/root/reference/train.py:521-535``): ``run_inference(..., training_mode=True)``
ends in ``vae.decode(latents / scaling_factor).sample.clamp(-1, 1)`` (``/root/reference/models/infer.py:121-123``) and the face loss
back-propagates through it into the last denoising step.  The VAE is frozen: only data gradients.

The mid-block attention is one head of width 512 - too wide for the flash kernels' register tiles - so, like the forward, it is composed
from GEMMs per image: S = Q K^T, P = softmax(S / sqrt(c)) (kept), O = P V; backward dV = P^T dO, dP = dO V^T,
dS = P (dP - rowsum(P dP)) / sqrt(c) (``pv_softmax_rows_backward``), dQ = dS K, dK = dS^T Q, with ``pv_transpose_f16`` producing the
transposed operands.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from .ops import ACT_NONE, ACT_SILU
from .tape import Tape, Var, conv3_dgrad_weight
from .vae import _conv3_w, _f32


def _res(tp: Tape, vae, m, x: Var, b, h, w) -> Var:
    g = vae.config.norm_num_groups
    hn = tp.groupnorm(x, _f32(m.norm1.weight), _f32(m.norm1.bias), batch=b, hw=h * w, eps=m.norm1.eps, act=ACT_SILU, groups=g)
    h1 = tp.conv3(hn, _conv3_w(m.conv1.weight), conv3_dgrad_weight(m.conv1.weight), bias=_f32(m.conv1.bias), batch=b, h=h, w=w)
    h2 = tp.groupnorm(h1, _f32(m.norm2.weight), _f32(m.norm2.bias), batch=b, hw=h * w, eps=m.norm2.eps, act=ACT_SILU, groups=g)
    sc = x
    if m.conv_shortcut is not None:
        sc = tp.linear(x, *tp.frozen(m.conv_shortcut.weight.reshape(m.conv_shortcut.out_channels, -1)), bias=_f32(m.conv_shortcut.bias),
                       rows_per_image=h * w)
    return tp.conv3(h2, _conv3_w(m.conv2.weight), conv3_dgrad_weight(m.conv2.weight), bias=_f32(m.conv2.bias), batch=b, h=h, w=w, residual=sc)


def _attn(tp: Tape, vae, m, x: Var, b, h, w) -> Var:
    n, c = h * w, m.to_q.in_features
    if n % 128:
        raise NotImplementedError("VAE attention on the training tape needs h * w % 128 == 0 (the key count is a GEMM N dimension)")
    rf, rb = tp.rf, tp.rb
    scale = c ** -0.5
    g = tp.groupnorm(x, _f32(m.group_norm.weight), _f32(m.group_norm.bias), batch=b, hw=n, eps=m.group_norm.eps, act=ACT_NONE,
                     groups=vae.config.norm_num_groups)
    q = tp.linear(g, *tp.frozen(m.to_q.weight), bias=_f32(m.to_q.bias), rows_per_image=n)
    k = tp.linear(g, *tp.frozen(m.to_k.weight), bias=_f32(m.to_k.bias), rows_per_image=n)
    v = tp.linear(g, *tp.frozen(m.to_v.weight), bias=_f32(m.to_v.bias), rows_per_image=n)
    o = rf.empty((b * n, c))
    probs = []
    for i in range(b):
        rows = slice(i * n, (i + 1) * n)
        s = rf.gemm(q.t[rows], rf.hold(k.t[rows]), splitk=0)                 # S = Q K^T   [n, n], kept as P
        rf.softmax_rows(s, scale=scale)
        vt = rf.transpose(v.t[rows])                                           # V^T [c, n]
        rf.gemm(s, vt, out=o[rows], splitk=0)                                  # O = P V
        probs.append(s)
    ov = Var(o, True)

    def bwd():
        if ov.g is None:
            return
        dq, dk, dv = rb.empty((b * n, c)), rb.empty((b * n, c)), rb.empty((b * n, c))
        for i in range(b):
            rows = slice(i * n, (i + 1) * n)
            do = ov.g[rows]
            rb.gemm(rb.transpose(probs[i]), rb.transpose(do), out=dv[rows], splitk=0)     # dV = P^T dO
            dp = rb.gemm(do, rb.hold(v.t[rows]), splitk=0)                                # dP = dO V^T
            rb.softmax_rows_backward(probs[i], dp, scale=scale)                           # dS (in place)
            rb.gemm(dp, rb.transpose(k.t[rows]), out=dq[rows], splitk=0)                  # dQ = dS K
            rb.gemm(rb.transpose(dp), rb.transpose(q.t[rows]), out=dk[rows], splitk=0)    # dK = dS^T Q
        tp._accum(q, dq)
        tp._accum(k, dk)
        tp._accum(v, dv)
    tp.back.append(bwd)
    return tp.linear(ov, *tp.frozen(m.to_out[0].weight), bias=_f32(m.to_out[0].bias), residual=x, rows_per_image=n)


def decode_on_tape(tp: Tape, vae, z: torch.Tensor, *, clamp=(-1.0, 1.0)):
    """``z``: fp32 (B, latent_channels, h, w) buffer holding ``latents / scaling_factor``.  Records the decoder forward; returns a
    namespace with ``img`` (fp32 (B, 3, 8h, 8w), clamped like infer.py:122), ``dimg`` (holder: set ``.g`` to the fp32 image gradient
    BEFORE ``tp.build_backward()``... i.e. by a closure pushed after this call) and ``dz`` (holder: ``.g`` = gradient w.r.t. ``z``)."""
    cfg, d = vae.config, vae.decoder
    rf, rb = tp.rf, tp.rb
    dev = z.device
    B, lc, h, w = z.shape
    dz, dimg = SimpleNamespace(g=None), SimpleNamespace(g=None)
    wpq = _f32(vae.post_quant_conv.weight.reshape(lc, lc))
    zq = rf.pointwise_nchw(z, wpq, _f32(vae.post_quant_conv.bias), batch=B, cin=lc, cout=lc, hw=h * w)
    c_in = d.conv_in.out_channels
    kin, kpad = lc * 9, (lc * 9 + 63) // 64 * 64
    cols = rf.im2col3x3(zq, batch=B, cin=lc, h=h, wd=w, kpad=kpad)
    w_in = torch.zeros(c_in, kpad, dtype=torch.float16, device=dev)
    w_in[:, :kin] = d.conv_in.weight.detach().reshape(c_in, kin).to(torch.float16)
    x = tp.linear(Var(cols, False), w_in, w_in.t().contiguous(), bias=_f32(d.conv_in.bias), rows_per_image=h * w, colstats=True)
    x.needs = True
    h0, w0 = h, w

    def head_bwd(x=x):
        if x.g is None:
            return
        # d/d(zq): conv_in's data gradient is a c_in -> 4 channel 3x3 conv of dX with the flipped filter (pv_conv_out), then post_quant_conv^T
        wdg = d.conv_in.weight.detach().flip(2, 3).permute(1, 2, 3, 0).reshape(lc, -1).to(torch.float16).contiguous()
        g = x.g if x.g.is_contiguous() else rb.add_rows(x.g, torch.zeros_like(x.t))
        dzq = rb.conv_out(g, wdg, None, batch=B, cin=c_in, h=h0, wd=w0, cout=lc)
        dz.g = rb.pointwise_nchw(dzq, wpq.t().contiguous(), None, batch=B, cin=lc, cout=lc, hw=h0 * w0).view(B, lc, h0, w0)
    tp.back.append(head_bwd)
    x = _res(tp, vae, d.mid_block.resnets[0], x, B, h, w)
    x = _attn(tp, vae, d.mid_block.attentions[0], x, B, h, w)
    x = _res(tp, vae, d.mid_block.resnets[1], x, B, h, w)
    for blk in d.up_blocks:
        for r in blk.resnets:
            x = _res(tp, vae, r, x, B, h, w)
        if blk.upsamplers is not None:
            conv = blk.upsamplers[0].conv
            x = tp.conv3(x, _conv3_w(conv.weight), conv3_dgrad_weight(conv.weight), bias=_f32(conv.bias), batch=B, h=h, w=w, upsample=1)
            h, w = 2 * h, 2 * w
    xn = tp.groupnorm(x, _f32(d.conv_norm_out.weight), _f32(d.conv_norm_out.bias), batch=B, hw=h * w, eps=d.conv_norm_out.eps, act=ACT_SILU,
                      groups=cfg.norm_num_groups)
    co, ci = d.conv_out.out_channels, d.conv_out.in_channels
    wo = d.conv_out.weight.detach().permute(0, 2, 3, 1).reshape(co, -1).to(torch.float16).contiguous()
    img = rf.conv_out(xn.t, wo, _f32(d.conv_out.bias), batch=B, cin=ci, h=h, wd=w, cout=co)
    if clamp is not None:
        rf.clamp_(img, clamp[0], clamp[1])
    hh, ww = h, w

    def tail_bwd():
        if dimg.g is None:
            return
        gimg = rb.clamp_mask(img, dimg.g, clamp[0], clamp[1]) if clamp is not None else dimg.g      # post-clamp values decide the mask
        kp = (co * 9 + 63) // 64 * 64
        dcols = rb.im2col3x3(gimg, batch=B, cin=co, h=hh, wd=ww, kpad=kp)
        wd_out = torch.zeros(ci, kp, dtype=torch.float16, device=dev)
        wd_out[:, :co * 9] = d.conv_out.weight.detach().flip(2, 3).permute(1, 0, 2, 3).reshape(ci, co * 9).to(torch.float16)
        xn.g = rb.gemm(dcols, wd_out, rows_per_image=hh * ww)
    tp.back.append(tail_bwd)
    return SimpleNamespace(img=img, dimg=dimg, dz=dz)
