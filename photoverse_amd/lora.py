"""Minimal LoRA support compatible with the checkpoint layout the reference writes through peft
(``/root/reference/models/modeling_utils.py:15-18,45-46,86-88``; ``train.py:348-354``): injected Linear layers are renamed
``<name>.base_layer`` and gain ``<name>.lora_A.default`` (r x in) and ``<name>.lora_B.default`` (out x r).
For the HIP inference path the low-rank update is merged into the packed fp16 weight (W + alpha/r * B @ A); the training plan
(``train.TrainStep``) does the same when ``lora_dropout == 0`` and runs the low-rank branch separately, with a device-side dropout on
its input, otherwise.
peft itself is not installable here ([EXT] peft==0.10.0), so ``LoraConfig`` mirrors only the fields the reference uses.
"""
from __future__ import annotations

from dataclasses import asdict, dataclass, field
from typing import List

import torch
import torch.nn as nn


@dataclass
class LoraConfig:
    r: int = 8
    lora_alpha: int = 8
    target_modules: List[str] = field(default_factory=lambda: ["attn2.to_q", "attn2.to_k", "attn2.to_v"])
    lora_dropout: float = 0.0
    init_lora_weights: str = "gaussian"

    def to_dict(self):
        return asdict(self)


class LoRALinear(nn.Module):
    def __init__(self, base: nn.Linear, r: int, alpha: float, dropout: float = 0.0):
        super().__init__()
        self.dropout_p = float(dropout)      # peft: result += B(A(dropout(x))) * scaling in train mode (identity at inference)
        self.base_layer = base
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, r, bias=False)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, base.out_features, bias=False)})
        nn.init.normal_(self.lora_A["default"].weight, std=1.0 / r)
        nn.init.zeros_(self.lora_B["default"].weight)
        self.scaling = alpha / r
        self.in_features, self.out_features = base.in_features, base.out_features

    @property
    def weight(self) -> torch.Tensor:
        """Merged weight (what the packed HIP GEMM consumes)."""
        return self.base_layer.weight + self.scaling * (self.lora_B["default"].weight @ self.lora_A["default"].weight)

    @property
    def bias(self):
        return self.base_layer.bias


def inject_adapter_in_model(config: LoraConfig, model: nn.Module) -> nn.Module:
    """Wrap every Linear whose qualified name ends with one of ``target_modules`` (peft's suffix matching)."""
    if isinstance(config, dict):
        config = LoraConfig(**{k: v for k, v in config.items() if k in LoraConfig.__dataclass_fields__})
    targets = []
    for name, mod in model.named_modules():
        if isinstance(mod, nn.Linear) and any(name.endswith(t) for t in config.target_modules) and not name.endswith("base_layer"):
            targets.append(name)
    for name in targets:
        parent_name, _, leaf = name.rpartition(".")
        parent = model.get_submodule(parent_name) if parent_name else model
        base = getattr(parent, leaf)
        wrapped = LoRALinear(base, config.r, config.lora_alpha, config.lora_dropout).to(base.weight.device)
        setattr(parent, leaf, wrapped)
    if hasattr(model, "repack"):
        model.repack()
    return model
