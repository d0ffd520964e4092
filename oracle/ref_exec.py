"""TEST INFRASTRUCTURE, BUILD CONTAINER ONLY: executes the reference's OWN function / class bodies.

Most reference modules do not import here: ``models/attention_processor.py`` and ``models/unet.py`` start with
``from diffusers ...`` (absent), ``models/clip.py`` imports a docstring constant transformers 5.x removed,
``models/arcface_resnet.py`` / ``models/loss.py`` pull ``gdown`` / ``cv2`` / ``facenet_pytorch`` through
``utils/arcface_utils.py``.  None of those imports is used by the arithmetic this build restates.  So instead of importing the
modules, ``load_reference_defs`` parses a reference file where it lies under ``/root/reference``, keeps ONLY the named top-level
``def`` / ``class`` statements (untouched - same AST nodes, same line numbers, compiled with the reference path as file name so
tracebacks point into the reference), and executes them in a namespace the caller supplies.  Every name the namespace binds that
is NOT the reference's own object is listed in ``STAND_INS`` below with the reason it is safe.

Nothing is copied into the repository: the source is read at fixture-generation time (``oracle/make_golden.py``) and only the
resulting tensors are committed under ``tests/golden/``.  ``/root/reference`` does not exist on the GPU box; importing this module
there raises.
"""
from __future__ import annotations

import ast
import os
import warnings
from typing import Dict, Iterable, List, Optional

REF_ROOT = "/root/reference"

#: names bound by the loaders below that are not the reference's own objects, and why that cannot change a result
STAND_INS = {
    "attention_processor.Attention": "type annotation only (attention_processor.py:60,247); bound to `object`",
    "attention_processor.IPAdapterMaskProcessor": "used only inside the `ip_adapter_masks is not None` branch (:359-390), which no "
                                                  "PhotoVerse caller reaches (infer.py:103-114, train.py:505-506); bound to a class whose "
                                                  "every attribute access raises",
    "attention_processor.deprecate": "diffusers.utils.deprecate on the bare-tensor path (:263-273) only emits a warning; bound to a "
                                     "function that emits a FutureWarning",
    "unet.AttnProcessor2_0 / AttnProcessor": "[EXT] diffusers stock processors installed on attn1 (unet.py:20-24); bound to the oracle's "
                                             "AttnProcessor2_0Ref (the only [EXT] piece in that function)",
    "clip decorators": "`@add_start_docstrings_to_model_forward` / `@replace_return_docstrings` (clip.py:27-28) only edit __doc__; "
                       "stripped",
    "clip.self.encoder": "transformers 5.x `CLIPEncoder.forward` takes ONE additive mask; the shim adds `causal_attention_mask` and "
                         "`attention_mask` (what 4.40's encoder layer does) and drops the output_* / return_dict flags",
    "arcface_resnet.download_arcface_pytorch": "only called with pretrained=True (:130-131); not bound - fixtures use pretrained=False",
    "loss.InceptionResnetV1 / cv2": "facenet variant (:24) and the __main__ demo only; not bound",
    "loss.FaceLoss._load_model": "the reference downloads ArcFace weights (:21-22 -> arcface_resnet.py:129-134); the fixture builds the "
                                 "same class with pretrained=False and seeded weights instead (subclass overriding this staticmethod)",
}


def _require_reference() -> None:
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("oracle.ref_exec needs /root/reference (build container only); GPU-box tests use the committed fixtures")


def load_reference_defs(relpath: str, names: Iterable[str], namespace: Dict[str, object],
                        strip_decorators: Iterable[str] = ()) -> Dict[str, object]:
    """Execute the top-level ``def`` / ``class`` statements called ``names`` of ``/root/reference/<relpath>`` in ``namespace``."""
    _require_reference()
    path = os.path.join(REF_ROOT, relpath)
    with open(path) as fh:
        tree = ast.parse(fh.read(), filename=path)
    names = list(names)
    picked: List[ast.stmt] = []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            if node.name in strip_decorators:
                node.decorator_list = []
            picked.append(node)
    missing = set(names) - {n.name for n in picked}
    if missing:
        raise KeyError(f"{relpath}: no top-level definition named {sorted(missing)}")
    code = compile(ast.Module(body=picked, type_ignores=[]), filename=path, mode="exec")
    exec(code, namespace)
    return namespace


class _Unreachable:
    """Stand-in for a name on a branch the fixtures must never reach."""

    def __init__(self, what):
        object.__setattr__(self, "_what", what)

    def __getattr__(self, item):
        raise AssertionError(f"{object.__getattribute__(self, '_what')}.{item} touched: the fixture left the reachable branches")


def _deprecate(*args, **kwargs):
    warnings.warn(str(args[2]) if len(args) > 2 else "deprecated", FutureWarning)


def reference_attention_processors():
    """``PhotoVerseAttnProcessor`` / ``PhotoVerseAttnProcessor2_0`` exactly as ``models/attention_processor.py:12-435`` defines them."""
    from typing import List as _List
    from typing import Optional as _Optional

    import torch
    import torch.nn.functional as F
    from torch import nn
    ns = {"torch": torch, "F": F, "nn": nn, "Optional": _Optional, "List": _List, "Attention": object,
          "IPAdapterMaskProcessor": _Unreachable("IPAdapterMaskProcessor"), "deprecate": _deprecate}
    load_reference_defs("models/attention_processor.py", ["PhotoVerseAttnProcessor", "PhotoVerseAttnProcessor2_0"], ns)
    return ns["PhotoVerseAttnProcessor"], ns["PhotoVerseAttnProcessor2_0"]


def reference_unet_helpers():
    """``set_visual_cross_attention_adapter`` / ``get_visual_cross_attention_values_norm`` of ``models/unet.py:8-47``."""
    import torch
    import torch.nn.functional as F

    from oracle.unet_ref import AttnProcessor2_0Ref
    legacy, sdpa = reference_attention_processors()
    ns = {"torch": torch, "F": F, "AttnProcessor2_0": AttnProcessor2_0Ref, "AttnProcessor": AttnProcessor2_0Ref,
          "PhotoVerseAttnProcessor2_0": sdpa, "PhotoVerseAttnProcessor": legacy}
    load_reference_defs("models/unet.py", ["set_visual_cross_attention_adapter", "get_visual_cross_attention_values_norm"], ns)
    return ns["set_visual_cross_attention_adapter"], ns["get_visual_cross_attention_values_norm"]


def reference_clip_functions():
    """``_inject_concept_embeddings`` and ``clip_text_transformer_forward`` of ``models/clip.py:17-102`` (decorators stripped)."""
    from typing import Optional as _Optional
    from typing import Tuple as _Tuple
    from typing import Union as _Union

    import torch
    from transformers.modeling_attn_mask_utils import _create_4d_causal_attention_mask, _prepare_4d_attention_mask
    from transformers.modeling_outputs import BaseModelOutputWithPooling
    ns = {"torch": torch, "Optional": _Optional, "Tuple": _Tuple, "Union": _Union,
          "_create_4d_causal_attention_mask": _create_4d_causal_attention_mask, "_prepare_4d_attention_mask": _prepare_4d_attention_mask,
          "BaseModelOutputWithPooling": BaseModelOutputWithPooling}
    load_reference_defs("models/clip.py", ["_inject_concept_embeddings", "clip_text_transformer_forward"], ns,
                        strip_decorators=["clip_text_transformer_forward"])
    return ns["_inject_concept_embeddings"], ns["clip_text_transformer_forward"]


class EncoderShim:
    """Adapts the transformers-4.40 call ``self.encoder(inputs_embeds=, attention_mask=, causal_attention_mask=, output_*=,
    return_dict=)`` (``clip.py:75-82``) to the installed 5.x ``CLIPEncoder`` (one additive mask, no flags)."""

    def __init__(self, encoder):
        self.encoder = encoder

    def __call__(self, inputs_embeds, attention_mask=None, causal_attention_mask=None, output_attentions=None,
                 output_hidden_states=None, return_dict=None):
        mask = causal_attention_mask if attention_mask is None else causal_attention_mask + attention_mask
        return self.encoder(inputs_embeds=inputs_embeds, attention_mask=mask)


class TextTransformerShim:
    """``self`` for the reference's ``clip_text_transformer_forward``: the installed ``CLIPTextModel``'s own submodules."""

    def __init__(self, hf_text_model):
        from types import SimpleNamespace
        self.config = SimpleNamespace(output_attentions=False, output_hidden_states=False, use_return_dict=True)
        self.embeddings = hf_text_model.embeddings
        self.encoder = EncoderShim(hf_text_model.encoder)
        self.final_layer_norm = hf_text_model.final_layer_norm


def reference_arcface():
    """``IRBlock`` / ``SEBlock`` / ``ResNetFace`` / ``ArcFaceResNet18`` of ``models/arcface_resnet.py:6-134`` and ``FaceLoss`` of
    ``models/loss.py:9-78`` (its ``_load_model`` overridden: see STAND_INS)."""
    import torch
    import torch.nn as nn
    from torch.nn import CosineEmbeddingLoss
    from torch.nn import functional as F
    ns = {"torch": torch, "nn": nn}
    load_reference_defs("models/arcface_resnet.py", ["conv3x3", "IRBlock", "SEBlock", "ResNetFace", "ArcFaceResNet18"], ns)
    ns2 = {"torch": torch, "F": F, "CosineEmbeddingLoss": CosineEmbeddingLoss, "ArcFaceResNet18": ns["ArcFaceResNet18"]}
    load_reference_defs("models/loss.py", ["FaceLoss"], ns2)
    base = ns2["FaceLoss"]

    class FaceLossNoDownload(base):
        @staticmethod
        def _load_model(model_name):
            assert model_name == "arcface"
            return ns["ArcFaceResNet18"](pretrained=False)

    return ns["ArcFaceResNet18"], FaceLossNoDownload


# ---------------------------------------------------------------------------------------------------------------------------------
# models/infer.py:7-123 and models/modeling_utils.py:13-95: the orchestration code of the hot path (pure Python over duck-typed objects)
# ---------------------------------------------------------------------------------------------------------------------------------
STAND_INS.update({
    "infer.DPMSolverMultistepScheduler": "[EXT] diffusers sampler (infer.py:1,39-40): bound to SchedulerStandIn, a thin adapter over "
                                         "oracle.scheduler_ref.DPMSolverMultistepRef exposing exactly what run_inference touches (from_config, "
                                         "set_timesteps, timesteps, init_noise_sigma, scale_model_input, step(...).prev_sample, add_noise); the "
                                         "adapter holds no arithmetic of its own and logs every call so the fixture can assert the call sequence",
    "infer.tqdm": "the installed tqdm package itself (4.67) - not a stand-in",
    "infer arguments": "unet / vae / image_encoder / text_encoder are the oracle's tiny [EXT] restatements with the REFERENCE processor class "
                       "installed by the REFERENCE set_visual_cross_attention_adapter; adapters are the real models.adapters.PhotoVerseAdapter; "
                       "tokenizer is a recording stub returning fixed ids (the real CLIPTokenizer has no vocabulary offline)",
    "modeling_utils.LoraConfig": "[EXT] peft.LoraConfig (modeling_utils.py:2,17,46): bound to LoraConfigStandIn, a dataclass with the fields "
                                 "train.py:348-354 sets and peft's to_dict(); carries no arithmetic",
    "modeling_utils.inject_adapter_in_model": "[EXT] peft (modeling_utils.py:18,88): bound to oracle.lora_ref.inject_adapter_in_model_ref (independent "
                                              "restatement of peft's published wrapper layout: base_layer / lora_A.default / lora_B.default)",
    "modeling_utils.accelerator": "save_progress only calls accelerator.unwrap_model(m) (:30-33): the fixture passes an object whose unwrap_model "
                                  "strips a DDP-like `.module` wrapper, which is what accelerate's does",
    "modeling_utils.*.from_pretrained": "load_models (:55-60) downloads six [EXT] models; each class is bound to a recording factory whose "
                                        "from_pretrained(path, subfolder=) returns the oracle's tiny restatement of that model; PhotoVerseAdapter is the real "
                                        "class, set_visual_cross_attention_adapter the reference's own function, patch_clip_text_transformer an identity "
                                        "that records the call (the by-class-name patch of clip.py:115-119 targets a transformers-4.40 class)",
})


class SchedulerStandIn:
    """Plays ``diffusers.DPMSolverMultistepScheduler`` for ``models/infer.py``; every method forwards to ``DPMSolverMultistepRef``."""
    log: list = []                       # class-level call log, cleared by the fixture generator before each reference call

    def __init__(self, config):
        from oracle.scheduler_ref import DPMSolverMultistepRef
        cfg = config if isinstance(config, dict) else dict(getattr(config, "__dict__", {}))
        self._ref = DPMSolverMultistepRef(cfg.get("num_train_timesteps", 1000), cfg.get("beta_start", 0.00085), cfg.get("beta_end", 0.012),
                                          cfg.get("steps_offset", 1))
        self.config = config

    @classmethod
    def from_config(cls, config):
        cls.log.append(("from_config",))
        return cls(config)

    def set_timesteps(self, n):
        type(self).log.append(("set_timesteps", int(n)))
        self._ref.set_timesteps(n)

    @property
    def timesteps(self):
        return self._ref.timesteps

    @property
    def init_noise_sigma(self):
        return self._ref.init_noise_sigma

    def scale_model_input(self, sample, t):
        type(self).log.append(("scale_model_input", int(t)))
        return self._ref.scale_model_input(sample, t)

    def step(self, model_output, t, sample):
        from types import SimpleNamespace
        type(self).log.append(("step", int(t)))
        return SimpleNamespace(prev_sample=self._ref.step(model_output, t, sample))

    def add_noise(self, original_samples, noise, timesteps):
        type(self).log.append(("add_noise", [int(v) for v in timesteps]))
        return self._ref.add_noise(original_samples, noise, timesteps)


def reference_run_inference():
    """``run_inference`` exactly as ``models/infer.py:7-123`` defines it (module-level names bound per STAND_INS)."""
    import torch
    from tqdm import tqdm
    ns = {"torch": torch, "tqdm": tqdm, "DPMSolverMultistepScheduler": SchedulerStandIn}
    load_reference_defs("models/infer.py", ["run_inference"], ns)
    return ns["run_inference"]


def LoraConfigStandIn(**kw):
    """peft.LoraConfig stand-in: the fields the reference sets (train.py:348-354) + ``to_dict`` (modeling_utils.py:46)."""
    from dataclasses import asdict, dataclass, field
    from typing import List as _List

    @dataclass
    class LoraConfig:
        r: int = 8
        lora_alpha: int = 8
        target_modules: _List[str] = field(default_factory=lambda: ["attn2.to_q", "attn2.to_k", "attn2.to_v"])
        lora_dropout: float = 0.0
        init_lora_weights: str = "gaussian"
        bias: str = "none"                      # train.py:352 passes bias="none"

        def to_dict(self):
            return asdict(self)

    return LoraConfig(**kw)


def _inject_stand_in(lora_config, model):
    from oracle.lora_ref import inject_adapter_in_model_ref
    return inject_adapter_in_model_ref(model, lora_config.r, lora_config.lora_alpha, lora_config.target_modules, lora_config.lora_dropout)


class AcceleratorStandIn:
    @staticmethod
    def unwrap_model(m):
        return getattr(m, "module", m) if type(m).__name__ == "DDPLike" else m


def reference_checkpoint_functions():
    """``load_photoverse_model`` / ``save_progress`` exactly as ``models/modeling_utils.py:13-50`` defines them."""
    import torch
    ns = {"torch": torch, "os": os, "LoraConfig": LoraConfigStandIn, "inject_adapter_in_model": _inject_stand_in}
    load_reference_defs("models/modeling_utils.py", ["load_photoverse_model", "save_progress"], ns)
    return ns["load_photoverse_model"], ns["save_progress"]


def reference_load_models(factories: Dict[str, object], call_log: list):
    """``load_models`` (``models/modeling_utils.py:53-95``) with the six ``from_pretrained`` classes bound to recording factories.
    ``factories``: class name -> zero-argument callable building the oracle's tiny restatement of that model."""
    import sys

    import torch
    sys.path.insert(0, REF_ROOT)
    try:
        from models.adapters import PhotoVerseAdapter          # the real class (imports without diffusers)
    finally:
        sys.path.pop(0)
    set_adapter, _ = reference_unet_helpers()

    def recording(name):
        class _Factory:
            @staticmethod
            def from_pretrained(path, subfolder=None):
                call_log.append((name, path, subfolder))
                return factories[name]()
        _Factory.__name__ = name
        return _Factory

    def patch_identity(text_encoder):
        call_log.append(("patch_clip_text_transformer", type(text_encoder).__name__, None))
        return text_encoder

    ns = {"torch": torch, "os": os, "LoraConfig": LoraConfigStandIn, "inject_adapter_in_model": _inject_stand_in,
          "PhotoVerseAdapter": PhotoVerseAdapter, "patch_clip_text_transformer": patch_identity,
          "set_visual_cross_attention_adapter": set_adapter}
    for name in ("CLIPTokenizer", "CLIPTextModel", "AutoencoderKL", "UNet2DConditionModel", "CLIPVisionModel", "DDPMScheduler"):
        ns[name] = recording(name)
    load_reference_defs("models/modeling_utils.py", ["load_photoverse_model", "save_progress", "load_models"], ns)
    return ns["load_models"]
