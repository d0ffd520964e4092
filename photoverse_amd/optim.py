"""AdamW + gradient-norm clipping on HIP kernels for the parameters PhotoVerse trains (``/root/reference/train.py:372-377`` optimizer,
``:538-541`` ``clip_grad_norm_(…, 1)`` per module, ``:545`` ``optimizer.step()``).

Same update rule as ``torch.optim.AdamW`` (decoupled weight decay, bias correction).  The clip coefficient and the loss-scale removal
are computed ON THE DEVICE (sum of squares by the deterministic reduction kernel -> ``pv_clip_coef`` -> read by ``pv_adamw_step``), so a
step needs no host synchronisation.  State (exp_avg, exp_avg_sq) is fp32; parameters must be fp32 CUDA tensors.
"""
from __future__ import annotations

from typing import Iterable, List, Sequence

import torch

from .ops import Recorder, _ptr


class AdamW:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32:
                raise RuntimeError("photoverse_amd.optim.AdamW updates fp32 parameters on a HIP device (no CPU path)")
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.state = {id(p): (torch.zeros_like(p), torch.zeros_like(p)) for p in self.params}

    # ---- torch.optim.AdamW's checkpoint format (modeling_utils.save_progress stores optimizer.state_dict(), modeling_utils.py:43-44) ----
    def state_dict(self):
        state = {}
        if self.step_count:
            for i, p in enumerate(self.params):
                m, v = self.state[id(p)]
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m, "exp_avg_sq": v}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": True,     # torch >= 2.6: AdamW is Adam with this flag; a group without it loads as coupled L2 decay
                 "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        group = sd["param_groups"][0]
        if len(group["params"]) != len(self.params):
            raise ValueError("loaded state dict has a different number of parameters")
        self.lr, self.betas, self.eps, self.weight_decay = group["lr"], tuple(group["betas"]), group["eps"], group["weight_decay"]
        steps = set()
        for i, p in enumerate(self.params):
            st = sd["state"].get(i)
            if st is None:
                continue
            m, v = self.state[id(p)]
            m.copy_(st["exp_avg"])
            v.copy_(st["exp_avg_sq"])
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("per-parameter step counts differ: this optimizer keeps one step counter")
        self.step_count = steps.pop() if steps else 0

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self, clip_groups: Sequence[Sequence[torch.nn.Parameter]] = (), max_norm: float = 1.0, grad_scale: float = 1.0):
        """One AdamW step.  ``clip_groups``: parameter groups each clipped to ``max_norm`` by its own total gradient norm (the
        reference clips text_adapter, image_adapter and unet parameters separately); parameters in no group are not clipped.
        ``grad_scale``: the loss scale the gradients carry (divided out).  Returns the device tensors of the group norms."""
        self.step_count += 1
        dev = self.params[0].device
        rec = Recorder(dev)
        coef_of = {}
        norms = []
        for group in clip_groups:
            gs = [p for p in group if p.grad is not None]
            if not gs:
                continue
            sq = rec.empty((len(gs),), torch.float32)
            for i, p in enumerate(gs):
                g = p.grad.contiguous().view(-1)
                rec.hold(g)
                rec.reduce_sumsq(g, out=sq[i:i + 1], scale=1.0 / (grad_scale * grad_scale))
            coef = rec.empty((2,), torch.float32)
            rec._add(rec.lib.pv_clip_coef, _ptr(sq), len(gs), float(max_norm), 1.0 / grad_scale, _ptr(coef))
            norms.append(coef)
            for p in gs:
                coef_of[id(p)] = coef
        plain = None
        if grad_scale != 1.0:
            plain = rec.hold(torch.full((1,), 1.0 / grad_scale, dtype=torch.float32, device=dev))
        for p in self.params:
            if p.grad is None:
                continue
            m, v = self.state[id(p)]
            g = rec.hold(p.grad.contiguous())
            cs = coef_of.get(id(p), plain)
            rec._add(rec.lib.pv_adamw_step, _ptr(p.data), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(self.lr), float(self.betas[0]),
                     float(self.betas[1]), float(self.eps), float(self.weight_decay), self.step_count, _ptr(cs))
        rec.run()
        return [c[1:2] for c in norms]
