"""Oracle: fp32 eager restatement of the SD-v1.5 ``UNet2DConditionModel`` forward.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.  **PARITY UNPINNED** against
the real diffusers model (third-party ``diffusers==0.27.2``,
``/root/reference/requirements.txt:1``, not vendored, not installable here).

The reference never defines the UNet itself; it obtains it with
``UNet2DConditionModel.from_pretrained(..., subfolder="unet")``
(``/root/reference/models/modeling_utils.py:57``) and calls it at
``/root/reference/models/infer.py:103-114`` and ``train.py:505-506``.  What is
restated here is the published SD-v1.5 architecture (SURVEY.md section 8a row
U) with diffusers' module / state-dict names, so that

* ``save_progress``'s key filter (``modeling_utils.py:34-37``) works unchanged,
* ``set_visual_cross_attention_adapter`` (``models/unet.py:8-35``) can walk
  ``unet.attn_processors`` / ``unet.config`` exactly as it does on diffusers,
* real SD-v1.5 weights could be loaded with ``load_state_dict``.

The PhotoVerse processor restated in ``PhotoVerseAttnProcessor2_0Ref`` follows
``/root/reference/models/attention_processor.py:245-435`` line by line for the
branches the reference's callers reach.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------
# attention (diffusers ``Attention`` + processors)
# --------------------------------------------------------------------------
class AttnProcessor2_0Ref:
    """Stock self-attention processor installed on ``attn1`` by
    ``/root/reference/models/unet.py:20-24`` ([EXT] diffusers ``AttnProcessor2_0``)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, **kw):
        batch_size = hidden_states.shape[0]
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)
        head_dim = key.shape[-1] // attn.heads
        query = query.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        key = key.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        value = value.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        hidden_states = F.scaled_dot_product_attention(query, key, value, dropout_p=0.0, is_causal=False)
        hidden_states = hidden_states.transpose(1, 2).reshape(batch_size, -1, attn.heads * head_dim)
        hidden_states = attn.to_out[0](hidden_states)
        hidden_states = attn.to_out[1](hidden_states)
        return hidden_states


class PhotoVerseAttnProcessor2_0Ref(nn.Module):
    """Restatement of ``PhotoVerseAttnProcessor2_0``
    (``/root/reference/models/attention_processor.py:221-435``; ``__init__`` of
    the base class at ``:27-56``).

    For SD-v1.5 ``attn2`` the spatial_norm / group_norm / norm_cross /
    attention_mask / residual branches are no-ops and no caller passes
    ``ip_adapter_masks`` (``infer.py:103-114``, ``train.py:505-506``), so only
    the mask-free branch ``:391-420`` is restated.
    """

    def __init__(self, hidden_size, cross_attention_dim=None, num_tokens=(5,), scale=2.0, fusion_rules=(1 / 3, 2 / 3)):
        super().__init__()
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        if not isinstance(num_tokens, (tuple, list)):
            num_tokens = [num_tokens]
        self.num_tokens = num_tokens
        # attention_processor.py:37-43
        if not isinstance(fusion_rules, tuple) or len(fusion_rules) != 2 or not all(isinstance(i, float) for i in fusion_rules):
            raise ValueError("`fusion_rules` should be a tuple of two floats.")
        self.fusion_rule1, self.fusion_rule2 = fusion_rules
        if self.fusion_rule1 + self.fusion_rule2 != 1:
            raise ValueError("Sum of the fusion rules should be equal to 1.")
        # attention_processor.py:45-49
        if not isinstance(scale, list):
            scale = [scale] * len(num_tokens)
        if len(scale) != len(num_tokens):
            raise ValueError("`scale` should be a list of integers with the same length as `num_tokens`.")
        self.scale = scale
        # attention_processor.py:51-56
        self.to_k_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False) for _ in range(len(num_tokens))])
        self.to_v_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False) for _ in range(len(num_tokens))])
        self.to_v_ip_norm = None
        #: test hook: when not None, replaces ``torch.rand(1).item()`` (``:414``)
        self.forced_fusion_seed: Optional[float] = None

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=2.0,
                 ip_adapter_masks=None):
        # attention_processor.py:258-262 (tuple convention; the deprecated bare-tensor split is :263-273)
        if isinstance(encoder_hidden_states, tuple):
            encoder_hidden_states, ip_hidden_states = encoder_hidden_states
            if not isinstance(ip_hidden_states, list):
                ip_hidden_states = [ip_hidden_states]
        else:
            end_pos = encoder_hidden_states.shape[1] - self.num_tokens[0]
            encoder_hidden_states, ip_hidden_states = (
                encoder_hidden_states[:, :end_pos, :],
                [encoder_hidden_states[:, end_pos:, :]],
            )
        if ip_adapter_masks is not None:
            raise NotImplementedError("ip_adapter_masks branch (:324-390) is dead for PhotoVerse callers")
        batch_size = encoder_hidden_states.shape[0]

        query = attn.to_q(hidden_states)                       # :297
        key = attn.to_k(encoder_hidden_states)                  # :304
        value = attn.to_v(encoder_hidden_states)                # :305
        inner_dim = key.shape[-1]
        head_dim = inner_dim // attn.heads
        query = query.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        key = key.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        value = value.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
        hidden_states = F.scaled_dot_product_attention(query, key, value, attn_mask=None, dropout_p=0.0, is_causal=False)  # :317
        hidden_states = hidden_states.transpose(1, 2).reshape(batch_size, -1, attn.heads * head_dim)

        for current_ip_hidden_states, scale, to_k_ip, to_v_ip in zip(ip_hidden_states, self.scale, self.to_k_ip, self.to_v_ip):
            ip_key = to_k_ip(current_ip_hidden_states)          # :392
            ip_value = to_v_ip(current_ip_hidden_states)        # :393
            ip_key = ip_key.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
            ip_value = ip_value.view(batch_size, -1, attn.heads, head_dim).transpose(1, 2)
            self.to_v_ip_norm = torch.norm(ip_value, dim=-1, keepdim=True)  # :397
            current_ip_hidden_states = F.scaled_dot_product_attention(query, ip_key, ip_value, attn_mask=None, dropout_p=0.0,
                                                                      is_causal=False)  # :400
            current_ip_hidden_states = current_ip_hidden_states.transpose(1, 2).reshape(batch_size, -1, attn.heads * head_dim)
            if not torch.is_grad_enabled():                      # :411-412
                hidden_states = hidden_states + current_ip_hidden_states
            else:                                                # :413-420
                seed = torch.rand(1).item() if self.forced_fusion_seed is None else self.forced_fusion_seed
                if seed < self.fusion_rule1:
                    hidden_states = scale * hidden_states
                elif seed > self.fusion_rule2:
                    hidden_states = scale * current_ip_hidden_states
                else:
                    hidden_states = hidden_states + current_ip_hidden_states

        hidden_states = attn.to_out[0](hidden_states)           # :423
        hidden_states = attn.to_out[1](hidden_states)           # :425
        return hidden_states


class AttentionRef(nn.Module):
    """[EXT] diffusers ``Attention`` as configured by SD-v1.5 transformer blocks
    (no bias on q/k/v, bias on ``to_out.0``, no norms, no residual)."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner_dim = heads * dim_head
        self.heads = heads
        self.inner_dim = inner_dim
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.scale = dim_head ** -0.5
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.to_q = nn.Linear(query_dim, inner_dim, bias=False)
        self.to_k = nn.Linear(self.cross_attention_dim, inner_dim, bias=False)
        self.to_v = nn.Linear(self.cross_attention_dim, inner_dim, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner_dim, query_dim, bias=True), nn.Dropout(0.0)])
        self.processor = AttnProcessor2_0Ref()

    def set_processor(self, processor):
        if hasattr(self, "processor") and isinstance(self.processor, nn.Module) and not isinstance(processor, nn.Module):
            self._modules.pop("processor")
        self.processor = processor

    def get_processor(self):
        return self.processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask, **kw)


class GEGLURef(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        hidden, gate = self.proj(x).chunk(2, dim=-1)
        return hidden * F.gelu(gate)


class FeedForwardRef(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        inner = dim * mult
        self.net = nn.ModuleList([GEGLURef(dim, inner), nn.Dropout(0.0), nn.Linear(inner, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlockRef(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = AttentionRef(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = AttentionRef(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForwardRef(dim)

    def forward(self, hidden_states, encoder_hidden_states=None):
        hidden_states = self.attn1(self.norm1(hidden_states)) + hidden_states
        hidden_states = self.attn2(self.norm2(hidden_states), encoder_hidden_states=encoder_hidden_states) + hidden_states
        hidden_states = self.ff(self.norm3(hidden_states)) + hidden_states
        return hidden_states


class Transformer2DModelRef(nn.Module):
    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups=32):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlockRef(inner, heads, dim_head, cross_attention_dim)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)

    def forward(self, hidden_states, encoder_hidden_states=None):
        b, _, h, w = hidden_states.shape
        residual = hidden_states
        hidden_states = self.proj_in(self.norm(hidden_states))
        inner = hidden_states.shape[1]
        hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(b, h * w, inner)
        for blk in self.transformer_blocks:
            hidden_states = blk(hidden_states, encoder_hidden_states=encoder_hidden_states)
        hidden_states = hidden_states.reshape(b, h, w, inner).permute(0, 3, 1, 2).contiguous()
        return self.proj_out(hidden_states) + residual


# --------------------------------------------------------------------------
# resnet / samplers
# --------------------------------------------------------------------------
class ResnetBlock2DRef(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.nonlinearity = nn.SiLU()
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(self, x, temb):
        h = self.conv1(self.nonlinearity(self.norm1(x)))
        h = h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None]
        h = self.conv2(self.dropout(self.nonlinearity(self.norm2(h))))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Downsample2DRef(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2DRef(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _DownBlock(nn.Module):
    def __init__(self, cin, cout, temb, layers, heads, xdim, groups, has_attn, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2DRef(cin if i == 0 else cout, cout, temb, groups) for i in range(layers)])
        if has_attn:
            self.attentions = nn.ModuleList([Transformer2DModelRef(heads, cout // heads, cout, xdim, groups) for _ in range(layers)])
        self.has_attn = has_attn
        self.downsamplers = nn.ModuleList([Downsample2DRef(cout)]) if add_down else None

    def forward(self, x, temb, ehs):
        outs = ()
        for i, res in enumerate(self.resnets):
            x = res(x, temb)
            if self.has_attn:
                x = self.attentions[i](x, encoder_hidden_states=ehs)
            outs += (x,)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs += (x,)
        return x, outs


class _MidBlock(nn.Module):
    def __init__(self, ch, temb, heads, xdim, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2DRef(ch, ch, temb, groups), ResnetBlock2DRef(ch, ch, temb, groups)])
        self.attentions = nn.ModuleList([Transformer2DModelRef(heads, ch // heads, ch, xdim, groups)])

    def forward(self, x, temb, ehs):
        x = self.resnets[0](x, temb)
        x = self.attentions[0](x, encoder_hidden_states=ehs)
        return self.resnets[1](x, temb)


class _UpBlock(nn.Module):
    def __init__(self, cin, cout, cprev, temb, layers, heads, xdim, groups, has_attn, add_up):
        super().__init__()
        resnets = []
        for i in range(layers):
            res_skip = cin if i == layers - 1 else cout
            res_in = cprev if i == 0 else cout
            resnets.append(ResnetBlock2DRef(res_in + res_skip, cout, temb, groups))
        self.resnets = nn.ModuleList(resnets)
        if has_attn:
            self.attentions = nn.ModuleList([Transformer2DModelRef(heads, cout // heads, cout, xdim, groups) for _ in range(layers)])
        self.has_attn = has_attn
        self.upsamplers = nn.ModuleList([Upsample2DRef(cout)]) if add_up else None

    def forward(self, x, skips, temb, ehs):
        for i, res in enumerate(self.resnets):
            x = torch.cat([x, skips[-1]], dim=1)
            skips = skips[:-1]
            x = res(x, temb)
            if self.has_attn:
                x = self.attentions[i](x, encoder_hidden_states=ehs)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


def timestep_embedding_ref(timesteps: torch.Tensor, dim: int = 320, max_period: float = 10000.0) -> torch.Tensor:
    """[EXT] diffusers ``get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)``."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half
    emb = timesteps[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)  # flipped: cos first


class _TimestepEmbedding(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


SD15_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5,
)

#: reduced config for fast CPU/GPU parity tests: same block kinds, head dims 40 and 80
TINY_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640), layers_per_block=1,
    down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
    attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5,
)


class UNet2DConditionModelRef(nn.Module):
    """SD-v1.5-shaped UNet; ``forward`` returns an object with ``.sample`` like
    diffusers' ``UNet2DConditionOutput`` (``infer.py:107``)."""

    def __init__(self, **overrides):
        super().__init__()
        cfg = dict(SD15_CONFIG)
        cfg.update(overrides)
        self.config = SimpleNamespace(**cfg)
        boc = tuple(cfg["block_out_channels"])
        heads, xdim, groups, layers = cfg["attention_head_dim"], cfg["cross_attention_dim"], cfg["norm_num_groups"], cfg["layers_per_block"]
        temb = boc[0] * 4
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)
        self.time_embedding = _TimestepEmbedding(boc[0], temb)
        self.down_blocks = nn.ModuleList()
        cout = boc[0]
        for i, kind in enumerate(cfg["down_block_types"]):
            cin, cout = cout, boc[i]
            self.down_blocks.append(_DownBlock(cin, cout, temb, layers, heads, xdim, groups, kind.startswith("CrossAttn"), i != len(boc) - 1))
        self.mid_block = _MidBlock(boc[-1], temb, heads, xdim, groups)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        cout = rev[0]
        for i, kind in enumerate(cfg["up_block_types"]):
            cprev, cout = cout, rev[i]
            cin = rev[min(i + 1, len(boc) - 1)]
            self.up_blocks.append(_UpBlock(cin, cout, cprev, temb, layers + 1, heads, xdim, groups, kind.startswith("CrossAttn"), i != len(boc) - 1))
        self.conv_norm_out = nn.GroupNorm(groups, boc[0], eps=cfg["norm_eps"])
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, padding=1)

    # -- diffusers attention-processor plumbing used by models/unet.py:8-47 --
    @property
    def attn_processors(self) -> Dict[str, object]:
        return {f"{n}.processor": m.get_processor() for n, m in self.named_modules() if isinstance(m, AttentionRef)}

    def set_attn_processor(self, processor):
        for n, m in self.named_modules():
            if isinstance(m, AttentionRef):
                m.set_processor(processor[f"{n}.processor"] if isinstance(processor, dict) else processor)

    def forward(self, sample, timestep, encoder_hidden_states=None):
        timesteps = timestep
        if not torch.is_tensor(timesteps):
            timesteps = torch.tensor([timesteps], dtype=torch.int64, device=sample.device)
        elif timesteps.ndim == 0:
            timesteps = timesteps[None].to(sample.device)
        timesteps = timesteps.expand(sample.shape[0])
        emb = self.time_embedding(timestep_embedding_ref(timesteps, self.config.block_out_channels[0]).to(sample.dtype))
        x = self.conv_in(sample)
        skips = (x,)
        for blk in self.down_blocks:
            x, outs = blk(x, emb, encoder_hidden_states)
            skips += outs
        x = self.mid_block(x, emb, encoder_hidden_states)
        for blk in self.up_blocks:
            n = len(blk.resnets)
            x = blk(x, skips[-n:], emb, encoder_hidden_states)
            skips = skips[:-n]
        x = self.conv_out(self.conv_act(self.conv_norm_out(x)))
        return SimpleNamespace(sample=x)


def set_visual_cross_attention_adapter_ref(unet: UNet2DConditionModelRef, num_tokens=(5,)):
    """Restatement of ``/root/reference/models/unet.py:8-35``."""
    procs = {}
    boc = unet.config.block_out_channels
    for name in unet.attn_processors.keys():
        cross_attention_dim = None if name.endswith("attn1.processor") else unet.config.cross_attention_dim
        if name.startswith("mid_block"):
            hidden_size = boc[-1]
        elif name.startswith("up_blocks"):
            hidden_size = list(reversed(boc))[int(name[len("up_blocks.")])]
        else:
            hidden_size = boc[int(name[len("down_blocks.")])]
        if cross_attention_dim is None:
            procs[name] = AttnProcessor2_0Ref()
        else:
            procs[name] = PhotoVerseAttnProcessor2_0Ref(hidden_size=hidden_size, cross_attention_dim=cross_attention_dim,
                                                        num_tokens=num_tokens)
    unet.set_attn_processor(procs)
    return unet


def get_visual_cross_attention_values_norm_ref(unet):
    """Restatement of ``/root/reference/models/unet.py:38-47``."""
    vals = [p.to_v_ip_norm for n, p in unet.attn_processors.items() if not n.endswith("attn1.processor")]
    out = torch.stack(vals, dim=1)
    return out.view(out.shape[0], -1)
