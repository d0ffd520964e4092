"""Oracle: fp32 eager restatement of the SD-v1.5 ``AutoencoderKL`` decode and encode paths.  TEST INFRASTRUCTURE, **PARITY UNPINNED**
([EXT] diffusers==0.27.2, not vendored / not installable; the reference loads it with
``AutoencoderKL.from_pretrained(..., subfolder="vae")``, ``/root/reference/models/modeling_utils.py:56`` and calls
``vae.decode(latents / vae.config.scaling_factor).sample.clamp(-1, 1)`` at ``/root/reference/models/infer.py:121-123``).

Restated from the public SD-v1.5 VAE definition with diffusers state-dict names: ``post_quant_conv`` (1x1, 4->4);
decoder ``conv_in`` 4->512; mid block = ResnetBlock, single-head self-attention over the H*W tokens (GroupNorm(32,1e-6),
to_q/k/v/out with bias, residual), ResnetBlock; four up blocks of 3 ResnetBlocks (512,512,256,128 channels; 1x1
``conv_shortcut`` when channels change) with nearest-x2 + 3x3 conv upsamplers on the first three; GroupNorm + SiLU +
``conv_out`` 128->3.  All GroupNorms: 32 groups, eps 1e-6.

``encode`` (``vae.encode(pixel_values).latent_dist.sample()``, ``/root/reference/models/infer.py:63`` for
``from_noised_image`` and ``/root/reference/train.py:471``): encoder ``conv_in`` 3->128; four down blocks of 2 ResnetBlocks
(128,256,512,512) with a Downsample2D on the first three = ``F.pad(x,(0,1,0,1))`` + Conv2d(3x3, stride 2, padding 0);
the same mid block; GroupNorm + SiLU + ``conv_out`` 512->8; ``quant_conv`` 1x1 8->8; the 8 channels are (mean, logvar)
of a diagonal Gaussian, logvar clamped to [-30, 20], ``sample() = mean + exp(0.5 logvar) * eps``.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Res(nn.Module):
    def __init__(self, cin, cout, groups=32):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class _Attn(nn.Module):
    def __init__(self, ch, groups=32):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, ch, eps=1e-6)
        self.to_q, self.to_k, self.to_v = nn.Linear(ch, ch), nn.Linear(ch, ch), nn.Linear(ch, ch)
        self.to_out = nn.ModuleList([nn.Linear(ch, ch), nn.Dropout(0.0)])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).view(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t)[:, None], self.to_k(t)[:, None], self.to_v(t)[:, None]     # one head of dim c
        o = F.scaled_dot_product_attention(q, k, v)[:, 0]
        o = self.to_out[0](o)
        return o.transpose(1, 2).reshape(b, c, h, w) + x


class _Mid(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(ch, ch, groups), _Res(ch, ch, groups)])
        self.attentions = nn.ModuleList([_Attn(ch, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Upsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Up(nn.Module):
    def __init__(self, cin, cout, layers, groups, add_up):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.upsamplers = nn.ModuleList([_Upsample(cout)]) if add_up else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.upsamplers is None else self.upsamplers[0](x)


class _Decoder(nn.Module):
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

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for u in self.up_blocks:
            x = u(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class _Downsample(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1)))


class _Down(nn.Module):
    def __init__(self, cin, cout, layers, groups, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([_Res(cin if i == 0 else cout, cout, groups) for i in range(layers)])
        self.downsamplers = nn.ModuleList([_Downsample(cout)]) if add_down else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.downsamplers is None else self.downsamplers[0](x)


class _Encoder(nn.Module):
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

    def forward(self, x):
        x = self.conv_in(x)
        for d in self.down_blocks:
            x = d(x)
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class DiagonalGaussianRef:
    def __init__(self, moments):
        self.mean, logvar = moments.chunk(2, dim=1)
        self.logvar = logvar.clamp(-30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, eps=None, generator=None):
        eps = torch.randn(self.mean.shape, generator=generator, dtype=self.mean.dtype) if eps is None else eps
        return self.mean + self.std * eps

    def mode(self):
        return self.mean


SD15_VAE_CONFIG = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                       norm_num_groups=32, scaling_factor=0.18215)
TINY_VAE_CONFIG = dict(latent_channels=4, out_channels=3, block_out_channels=(64, 128), layers_per_block=1, norm_num_groups=32,
                       scaling_factor=0.18215)


class AutoencoderKLDecoderRef(nn.Module):
    def __init__(self, **overrides):
        super().__init__()
        cfg = dict(SD15_VAE_CONFIG)
        cfg.update(overrides)
        self.config = SimpleNamespace(**cfg)
        self.post_quant_conv = nn.Conv2d(cfg["latent_channels"], cfg["latent_channels"], 1)
        self.decoder = _Decoder(cfg["latent_channels"], cfg["out_channels"], tuple(cfg["block_out_channels"]), cfg["layers_per_block"],
                                cfg["norm_num_groups"])
        if cfg.get("with_encoder", False):
            self.encoder = _Encoder(cfg.get("in_channels", 3), cfg["latent_channels"], tuple(cfg["block_out_channels"]),
                                    cfg["layers_per_block"], cfg["norm_num_groups"])
            self.quant_conv = nn.Conv2d(2 * cfg["latent_channels"], 2 * cfg["latent_channels"], 1)

    def encode(self, x):
        return SimpleNamespace(latent_dist=DiagonalGaussianRef(self.quant_conv(self.encoder(x))))

    def decode(self, z):
        return SimpleNamespace(sample=self.decoder(self.post_quant_conv(z)))
