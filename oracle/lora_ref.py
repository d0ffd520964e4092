"""TEST INFRASTRUCTURE (oracle): independent restatement of the peft LoRA ``Linear`` the reference trains through.

[EXT] ``peft==0.10.0`` (``/root/reference/requirements.txt:3``; not installable here, parity unpinned against the package):
``train.py:348-354`` builds ``LoraConfig(r, lora_alpha, target_modules=["attn2.to_q","attn2.to_k","attn2.to_v"] (+ "attn2.to_out.0"),
lora_dropout, init_lora_weights="gaussian")`` and ``modeling_utils.py:86-88`` calls ``inject_adapter_in_model(lora_config, unet)``.
peft's published behaviour, restated WITHOUT looking at ``photoverse_amd/lora.py``:

* a targeted ``nn.Linear`` called ``<name>`` becomes a wrapper with children ``base_layer`` (the original Linear),
  ``lora_dropout.default`` (``nn.Dropout(p)`` if p > 0 else ``nn.Identity``), ``lora_A.default = Linear(in, r, bias=False)``,
  ``lora_B.default = Linear(r, out, bias=False)``; ``scaling = lora_alpha / r``;
* ``init_lora_weights="gaussian"``: ``A ~ N(0, (1/r)^2)``, ``B = 0``;
* forward (un-merged): ``result = base_layer(x); result = result + lora_B(lora_A(dropout(x))) * scaling``;
* target matching: a module key matches when it equals a target or ends with ``"." + target``.

``dropout_hook`` lets a test impose the mask the device drew (the HIP path uses a counter-based generator; torch's CPU bit stream
cannot be reproduced on the device, so the comparison fixes the mask and checks everything else).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn as nn


class LoraLinearRef(nn.Module):
    def __init__(self, base_layer: nn.Linear, r: int, lora_alpha: float, lora_dropout: float = 0.0):
        super().__init__()
        self.base_layer = base_layer
        self.in_features, self.out_features = base_layer.in_features, base_layer.out_features
        self.r = r
        self.scaling = lora_alpha / r
        self.lora_dropout = nn.ModuleDict({"default": nn.Dropout(p=lora_dropout) if lora_dropout > 0.0 else nn.Identity()})
        self.lora_A = nn.ModuleDict({"default": nn.Linear(self.in_features, r, bias=False)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, self.out_features, bias=False)})
        nn.init.normal_(self.lora_A["default"].weight, std=1 / r)
        nn.init.zeros_(self.lora_B["default"].weight)
        #: test hook: ``hook(module, x) -> dropped x`` replaces ``lora_dropout`` when set
        self.dropout_hook: Optional[Callable] = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        result = self.base_layer(x)
        xd = self.dropout_hook(self, x) if self.dropout_hook is not None else self.lora_dropout["default"](x)
        return result + self.lora_B["default"](self.lora_A["default"](xd)) * self.scaling


def inject_adapter_in_model_ref(model: nn.Module, r: int, lora_alpha: float, target_modules, lora_dropout: float = 0.0) -> nn.Module:
    keys = [k for k, m in model.named_modules() if isinstance(m, nn.Linear) and any(k == t or k.endswith("." + t) for t in target_modules)]
    for key in keys:
        parent_name, _, leaf = key.rpartition(".")
        parent = model.get_submodule(parent_name) if parent_name else model
        setattr(parent, leaf, LoraLinearRef(getattr(parent, leaf), r, lora_alpha, lora_dropout))
    return model
