"""AdamW + gradient-norm clipping on HIP kernels for the parameters PhotoVerse trains (``/root/reference/train.py:372-377`` optimizer,
``:538-541`` ``clip_grad_norm_(…, 1)`` per module, ``:545`` ``optimizer.step()``).

Same update rule as ``torch.optim.AdamW`` (decoupled weight decay, bias correction).  The clip coefficient and the loss-scale removal
are computed ON THE DEVICE (``pv_sumsq_multi`` -> ``pv_clip_coef_groups`` -> read by ``pv_adamw_multi``: three multi-tensor launches over all
~220 parameter tensors, fixed summation order), so a step needs no host synchronisation.  State (exp_avg, exp_avg_sq) is fp32; parameters must be fp32 CUDA tensors.

Overflow guard: the training plans carry fp16 gradients under a static loss scale; the fp32 reference cannot overflow there.  When a
clip group's gradient norm is not finite the WHOLE step is skipped on the device (parameters - grouped or not -, moments and the
bias-correction step stay untouched), as ``torch.cuda.amp.GradScaler`` does; ``skipped_steps`` / ``applied_steps`` read the device counters.
The guard needs at least one clip group (the norm is what detects the overflow); a step without groups is unguarded.
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
        self.step_count = 0                # calls of step(); the bias correction uses the device's count of APPLIED steps
        self.state = {id(p): (torch.zeros_like(p), torch.zeros_like(p)) for p in self.params}
        self._counters = torch.zeros(2, dtype=torch.int32, device=self.params[0].device)     # [applied, skipped] (written by pv_clip_coef_groups)

    @property
    def applied_steps(self) -> int:
        """Steps whose update was applied (host sync).  Equals ``step_count`` unless gradients overflowed."""
        return int(self._counters[0].item()) if self._guarded else self.step_count

    @property
    def skipped_steps(self) -> int:
        return int(self._counters[1].item())

    _guarded = False                       # becomes True once a step ran with clip groups (the guard lives in the clip-coefficient kernel)

    # ---- torch.optim.AdamW's checkpoint format (modeling_utils.save_progress stores optimizer.state_dict(), modeling_utils.py:43-44) ----
    def state_dict(self):
        state = {}
        if self.step_count:
            applied = self.applied_steps
            for i, p in enumerate(self.params):
                m, v = self.state[id(p)]
                state[i] = {"step": torch.tensor(float(applied)), "exp_avg": m, "exp_avg_sq": v}
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
        self._counters.zero_()
        self._counters[0] = self.step_count

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None

    CHUNK = 32768      # elements per workgroup of the multi-tensor launches

    @torch.no_grad()
    def step(self, clip_groups: Sequence[Sequence[torch.nn.Parameter]] = (), max_norm: float = 1.0, grad_scale: float = 1.0):
        """One AdamW step.  ``clip_groups``: parameter groups each clipped to ``max_norm`` by its own total gradient norm (the
        reference clips text_adapter, image_adapter and unet parameters separately); parameters in no group are not clipped.
        ``grad_scale``: the loss scale the gradients carry (divided out).  Returns the device tensors of the group norms.

        Three launches whatever the number of tensors: the squared gradient norms (``pv_sumsq_multi``), the per-group clip coefficient
        times 1 / grad_scale (``pv_clip_coef_groups``, device memory) and the update (``pv_adamw_multi``) - no host synchronisation."""
        import numpy as np
        self.step_count += 1
        dev = self.params[0].device
        gid = {}
        for gi, group in enumerate(clip_groups):
            for p in group:
                gid[id(p)] = gi
        n_groups = len(clip_groups)
        active = [p for p in self.params if p.grad is not None]
        if not active:
            return []
        active.sort(key=lambda p: gid.get(id(p), n_groups))                     # stable: every group's tensors (and blocks) are contiguous
        grads = [p.grad if (p.grad.is_contiguous() and p.grad.dtype == torch.float32) else p.grad.float().contiguous() for p in active]
        used = sorted({gid[id(p)] for p in active if id(p) in gid})
        slot = {g: i for i, g in enumerate(used)}                               # groups without gradients are skipped
        key = (tuple((id(p), g.data_ptr()) for p, g in zip(active, grads)), tuple(gid.get(id(p), -1) for p in active), float(grad_scale))
        if getattr(self, "_mt_key", None) != key:
            # one {coefficient, norm} row per clip group + one trailing row for tensors in NO group: pv_clip_coef_groups sets it to 1 / grad_scale,
            # or to -1 when any group overflowed, so that ungrouped tensors are skipped together with the grouped ones
            coef = torch.zeros((len(used) + 1, 2), dtype=torch.float32, device=dev)
            plain = torch.full((1,), 1.0 / grad_scale, dtype=torch.float32, device=dev)
            entries = np.zeros((len(active), 6), dtype=np.int64)
            blk_t, blk_c, starts = [], [], [0] * (len(used) + 1)
            n_grouped = 0
            for t, (p, g) in enumerate(zip(active, grads)):
                m, v = self.state[id(p)]
                if id(p) in gid:
                    gs = coef.data_ptr() + 8 * slot[gid[id(p)]]
                elif used:
                    gs = coef.data_ptr() + 8 * len(used)                       # guarded step: the shared "ungrouped" row
                else:
                    gs = plain.data_ptr() if grad_scale != 1.0 else 0
                entries[t] = (p.data.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), gs, p.numel())
                nb = (p.numel() + self.CHUNK - 1) // self.CHUNK
                blk_t += [t] * nb
                blk_c += list(range(nb))
                if id(p) in gid:
                    n_grouped += nb
                    starts[slot[gid[id(p)]] + 1] = n_grouped
            self._mt = dict(coef=coef, plain=plain, entries=torch.from_numpy(entries).to(dev), blk_t=torch.tensor(blk_t, dtype=torch.int32, device=dev),
                            blk_c=torch.tensor(blk_c, dtype=torch.int32, device=dev), starts=torch.tensor(starts, dtype=torch.int32, device=dev),
                            partial=torch.zeros((max(n_grouped, 1),), dtype=torch.float32, device=dev), n_grouped=n_grouped, n_blocks=len(blk_t))
            self._mt_key = key
        mt = self._mt
        mt["grads"] = grads                                                    # keep the gradient buffers alive until the launches ran
        rec = Recorder(dev)
        if used:
            if not self._guarded:
                # first guarded step: the device counter takes over the bias-correction count - start it from the steps applied so far
                self._counters[0] = self.step_count - 1
            rec._add(rec.lib.pv_sumsq_multi, _ptr(mt["entries"]), _ptr(mt["blk_t"]), _ptr(mt["blk_c"]), mt["n_grouped"], self.CHUNK, _ptr(mt["partial"]))
            rec._add(rec.lib.pv_clip_coef_groups, _ptr(mt["partial"]), _ptr(mt["starts"]), len(used), float(max_norm), 1.0 / grad_scale, _ptr(mt["coef"]),
                     _ptr(self._counters))
            self._guarded = True
        elif self._guarded:
            self._counters[0] += 1                                               # keep the applied-step count moving without clip groups
        rec._add(rec.lib.pv_adamw_multi, _ptr(mt["entries"]), _ptr(mt["blk_t"]), _ptr(mt["blk_c"]), mt["n_blocks"], self.CHUNK, float(self.lr),
                 float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.weight_decay), self.step_count,
                 _ptr(self._counters) if self._guarded else 0)
        rec.run()
        return [mt["coef"][slot[g], 1:2] for g in used]
