"""A static forward + backward launch plan ("tape") over the HIP kernels.

The inference engines record ONE list of launches for a fixed shape and replay it (``ops.Recorder``).  Training does the same
with two lists: every forward helper below appends its launches to ``rf`` and pushes a closure that, called in reverse order
at BUILD time, appends the matching backward launches to ``rb``.  Nothing is traced and nothing runs while the plan is built;
a training step is ``rf.run(); rb.run()`` - two replays of fixed launch lists over fixed buffers (HIP-graph capturable like
the denoise loop), not a dynamic autograd graph.

Conventions: activations and their gradients are fp16 row matrices ``[M, C]`` (NHWC); a gradient may be a strided column view
of a wider buffer.  Gradients carry the loss scale ``S`` (they are scaled once, at the loss) and are accumulated functionally:
a tensor with several consumers gets ``g_new = contribution + g_old`` written to a fresh buffer, fused into the producing
kernel's epilogue (GEMM ``residual``, GroupNorm ``add``) where there is one.  Parameter gradients are fp32.

Data gradients of Linear / 3x3 conv layers are the forward MFMA GEMM on transposed / tap-flipped weights; weight gradients
are the same GEMM on transposed operands (``Recorder.wgrad``).  Reference: the training step these plans implement is
``/root/reference/train.py:466-545``.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from .ops import ACT_NONE, Recorder, _ptr


class Var:
    """A forward buffer, its gradient buffer (set while the backward plan is built) and whether a gradient is needed."""
    __slots__ = ("t", "g", "needs")

    def __init__(self, t: torch.Tensor, needs: bool = False):
        self.t, self.g, self.needs = t, None, needs


def conv3_dgrad_weight(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> fp16 [Cin, 9 * Cout]: the data gradient of a 3x3 / pad 1 convolution is the same convolution of dY
    with the taps flipped and the channel roles swapped."""
    return w.detach().flip(2, 3).permute(1, 2, 3, 0).reshape(w.shape[1], -1).to(torch.float16).contiguous()


class Tape:
    def __init__(self, device, grad_scale: float = 1.0):
        self.rf = Recorder(device)
        self.rb = Recorder(device)
        self.device = self.rf.device
        self.S = float(grad_scale)
        self.back: List[Callable[[], None]] = []
        self.refresh: List[tuple] = []          # (fp16 destination, fn() -> current fp32 / fp16 source): trainable weights, re-packed per step
        self.finalize: List[Callable[[], None]] = []   # after rb.run(): hand parameter gradients to the nn.Parameters
        self.packed: List[tuple] = []           # (fp32 master parameter, scale, fp16 block, transposed fp16 block): one pv_pack_weights launch per step
        self._pack_key = None

    # ------------------------------------------------------------------ weights
    def frozen(self, w: torch.Tensor) -> tuple:
        w16 = w.detach().to(torch.float16).contiguous()
        return w16, w16.t().contiguous()

    def trainable(self, fn: Callable[[], torch.Tensor]) -> tuple:
        """(w16, wT16) buffers re-filled from ``fn()`` (the current master weight, [N, K]) by ``load_weights`` before every step."""
        src = fn().detach()
        w16 = src.to(torch.float16).contiguous()
        wT = w16.t().contiguous()
        self.refresh.append((w16, wT, fn))
        return w16, wT

    def trainable_blocks(self, rows: int, cols: int, blocks) -> tuple:
        """(w16, wT16) of a [rows, cols] weight assembled from fp32 master PARAMETERS: ``blocks`` = [(param [r, c], row0, col0, scale)], zero
        elsewhere (block-diagonal LoRA factors, stacked projections, a single Linear weight).  All blocks of all such weights are re-made
        from the masters by ONE ``pv_pack_weights`` launch in ``load_weights`` - no per-tensor cast / transpose launches."""
        w16 = torch.zeros(rows, cols, dtype=torch.float16, device=self.device)
        wT = torch.zeros(cols, rows, dtype=torch.float16, device=self.device)
        for prm, r0, c0, scale in blocks:
            if prm.dim() != 2 or prm.dtype != torch.float32 or not prm.is_contiguous():
                raise TypeError("trainable_blocks takes contiguous fp32 [rows, cols] master weights")
            pr, pc = prm.shape
            if r0 + pr > rows or c0 + pc > cols:
                raise ValueError("block outside the weight")
            self.packed.append((prm, float(scale), w16[r0:r0 + pr, c0:c0 + pc], wT[c0:c0 + pc, r0:r0 + pr]))
        self._pack_key = None
        return w16, wT

    def _pack_table(self):
        import numpy as np
        key = tuple(prm.data_ptr() for prm, _, _, _ in self.packed)
        if key == self._pack_key:
            return
        ent = np.zeros((len(self.packed), 9), dtype=np.int64)
        blk_e, blk_t = [], []
        for i, (prm, scale, d, dT) in enumerate(self.packed):
            pr, pc = prm.shape
            ent[i] = (prm.data_ptr(), pc, pr, pc, int(np.float32(scale).view(np.int32)) & 0xFFFFFFFF, d.data_ptr(), d.stride(0), dT.data_ptr(), dT.stride(0))
            nt = ((pr + 31) // 32) * ((pc + 31) // 32)
            blk_e += [i] * nt
            blk_t += list(range(nt))
        dev = self.device
        self._pack = (torch.from_numpy(ent).to(dev), torch.tensor(blk_e, dtype=torch.int32, device=dev), torch.tensor(blk_t, dtype=torch.int32, device=dev))
        self.rw = Recorder(dev)
        self.rw._add(self.rw.lib.pv_pack_weights, _ptr(self._pack[0]), _ptr(self._pack[1]), _ptr(self._pack[2]), len(blk_e))
        self._pack_key = key

    @torch.no_grad()
    def load_weights(self):
        if self.packed:
            self._pack_table()                               # rebuilt only when a master moved (``.to()``, a replaced ``.data``)
            self.rw.run()
        for w16, wT, fn in self.refresh:                     # weights computed on the host side (merged LoRA: W + s B A)
            src = fn().detach()
            w16.copy_(src)
            wT.copy_(src.t())

    # ------------------------------------------------------------------ plumbing
    def _accum(self, var: Var, g: torch.Tensor):
        if not var.needs:
            return
        var.g = g if var.g is None else self.rb.add_rows(var.g, g)

    def build_backward(self):
        for fn in reversed(self.back):
            fn()
        self.back = []

    # ------------------------------------------------------------------ ops
    def linear(self, x: Var, w16, wT16, *, bias=None, x1: Optional[Var] = None, residual: Optional[Var] = None, rows_per_image=None,
               colstats=False, on_wgrad=None, on_bgrad=None, wgrad_scale=None) -> Var:
        y = self.rf.gemm(x.t, w16, a1=None if x1 is None else x1.t, bias=bias, residual=None if residual is None else residual.t,
                         rows_per_image=rows_per_image, colstats=colstats)
        out = Var(y, x.needs or (x1 is not None and x1.needs) or (residual is not None and residual.needs) or on_wgrad is not None
                  or on_bgrad is not None)

        def bwd():
            dy = out.g
            if dy is None:
                return
            if residual is not None:
                self._accum(residual, dy)
            if on_wgrad is not None:
                assert x1 is None
                on_wgrad(self.rb.wgrad(dy, x.t, scale=wgrad_scale))
            if on_bgrad is not None:
                on_bgrad(self.rb.colsum(dy))
            if x1 is None:
                if x.needs:
                    x.g = self.rb.gemm(dy, wT16, residual=x.g, rows_per_image=rows_per_image)
            elif x.needs or x1.needs:
                dcat = self.rb.gemm(dy, wT16, rows_per_image=rows_per_image)
                c0 = x.t.shape[1]
                self._accum(x, dcat[:, :c0])
                self._accum(x1, dcat[:, c0:])
        self.back.append(bwd)
        return out

    def conv3(self, x: Var, w16, wd16, *, bias, batch, h, w, rowadd=None, rowadd_ld=0, residual: Optional[Var] = None, stride=1, upsample=0,
              colstats=True) -> Var:
        """3x3 / pad 1 convolution over NHWC rows; ``stride`` 2 = Downsample2D, ``upsample`` 1 = Upsample2D (nearest x2 fused into the
        gather).  ``wd16`` = ``conv3_dgrad_weight`` of the same filter."""
        ho, wo = (h // 2, w // 2) if stride == 2 else ((h * 2, w * 2) if upsample else (h, w))
        geo = dict(batch=batch, hin=h, win=w, hout=ho, wout=wo, stride=stride, upsample=upsample)
        y = self.rf.gemm(x.t, w16, bias=bias, rowadd=rowadd, rowadd_ld=rowadd_ld, residual=None if residual is None else residual.t, conv=geo,
                         colstats=colstats)
        out = Var(y, x.needs or (residual is not None and residual.needs))

        def bwd():
            dy = out.g
            if dy is None:
                return
            if residual is not None:
                self._accum(residual, dy)
            if not x.needs:
                return
            if stride == 2:
                z = self.rb.dilate2x(dy, batch=batch, h=ho, w=wo)
                x.g = self.rb.gemm(z, wd16, conv=dict(batch=batch, hin=h, win=w, hout=h, wout=w), residual=x.g)
            elif upsample:
                du = self.rb.gemm(dy, wd16, conv=dict(batch=batch, hin=ho, win=wo, hout=ho, wout=wo))
                x.g = self.rb.pool2x_sum(du, batch=batch, h=h, w=w, add=x.g)
            else:
                x.g = self.rb.gemm(dy, wd16, conv=dict(batch=batch, hin=h, win=w, hout=h, wout=w), residual=x.g)
        self.back.append(bwd)
        return out

    def groupnorm(self, x: Var, gamma, beta, *, batch, hw, x1: Optional[Var] = None, eps=1e-5, act=ACT_NONE, groups=32) -> Var:
        y, stats = self.rf.groupnorm(x.t, gamma, beta, batch=batch, hw=hw, x1=None if x1 is None else x1.t, eps=eps, act=act, groups=groups,
                                     return_stats=True)
        out = Var(y, x.needs or (x1 is not None and x1.needs))

        def bwd():
            if out.g is None or not out.needs:
                return
            dx0, dx1 = self.rb.groupnorm_backward(x.t, out.g, stats, gamma, beta, batch=batch, hw=hw, x1=None if x1 is None else x1.t, act=act,
                                                  groups=groups, add0=x.g, add1=None if x1 is None else x1.g, want0=x.needs,
                                                  want1=x1 is not None and x1.needs)
            if x.needs:
                x.g = dx0
            if x1 is not None and x1.needs:
                x1.g = dx1
        self.back.append(bwd)
        return out

    def layernorm(self, x: Var, gamma, beta, *, eps=1e-5, act=ACT_NONE, on_affine=None, mean_group: int = 1) -> Var:
        """LayerNorm (+ activation).  ``mean_group`` = T > 1 additionally averages the rows 1..T-1 of every group of T rows
        (adapters.py:36: the patch-token mean, taken before the last Linear - a mean commutes with it) and returns [rows / T, cols]."""
        y = self.rf.layernorm(x.t, gamma, beta, eps=eps, act=act)
        if mean_group > 1:
            y = self.rf.rows_mean(y[1:], groups=x.t.shape[0] // mean_group, count=mean_group - 1, group_rows=mean_group)
        out = Var(y, x.needs or on_affine is not None)

        def bwd():
            if out.g is None:
                return
            kw = dict(dy_group=mean_group, dy_skip=1, dy_scale=1.0 / (mean_group - 1)) if mean_group > 1 else {}
            fuse = x.needs and x.g is not None and x.g.shape == x.t.shape      # the accumulation rides in the same launch
            dx, dgb = self.rb.layernorm_backward(x.t, out.g, gamma, beta, eps=eps, act=act, want_affine=on_affine is not None,
                                                 add=x.g if fuse else None, **kw)
            if on_affine is not None:
                on_affine(dgb)
            if fuse:
                x.g = dx
            else:
                self._accum(x, dx)
        self.back.append(bwd)
        return out

    def ln_linear(self, x: Var, gamma, beta, w16, wT16, *, eps=1e-5, bias=None, rows_per_image=None) -> Var:
        """``Linear(LayerNorm(x))`` with a FROZEN Linear: one row-owning launch where pv_row_gemm takes the shape (K = 320: the 64 x 64-level blocks'
        norm1 -> [to_q; to_k; to_v] and norm3 -> ff.net[0].proj), else LayerNorm + GEMM.  The normalised rows are never written: the backward needs
        only x (layernorm_backward recomputes the statistics) and the unfolded weight."""
        K, N = x.t.shape[1], w16.shape[0]
        if not (Recorder.row_gemm_supported(K, N) and x.t.shape[0] >= 4096 and x.t.is_contiguous()):
            return self.linear(self.layernorm(x, gamma, beta, eps=eps), w16, wT16, bias=bias, rows_per_image=rows_per_image)
        y = self.rf.row_gemm(x.t, w16, bias=bias, ln_gamma=gamma, ln_beta=beta, ln_eps=eps)
        out = Var(y, x.needs)

        def bwd():
            if out.g is None or not x.needs:
                return
            dn = self.rb.gemm(out.g, wT16, rows_per_image=rows_per_image)
            dx, _ = self.rb.layernorm_backward(x.t, dn, gamma, beta, eps=eps, want_affine=False, add=x.g)
            x.g = dx
        self.back.append(bwd)
        return out

    def self_attention(self, qkv: Var, *, batch, heads, n, d, causal=False) -> Var:
        C = heads * d
        lse = self.rf.empty((batch, heads, n), torch.float32)
        q, k, v = qkv.t[:, :C], qkv.t[:, C:2 * C], qkv.t[:, 2 * C:]
        o = self.rf.attention(q, k, v, batch=batch, heads=heads, nq=n, nk=n, d=d, causal=causal, lse=lse)
        out = Var(o, qkv.needs)

        def bwd():
            if out.g is None or not qkv.needs:
                return
            dqkv = self.rb.empty((batch * n, 3 * C))
            self.rb.attention_backward(q, k, v, o, out.g, lse, batch=batch, heads=heads, nq=n, nk=n, d=d, causal=causal, dq=dqkv[:, :C],
                                       dk=dqkv[:, C:2 * C], dv=dqkv[:, 2 * C:])
            self._accum(qkv, dqkv)
        self.back.append(bwd)
        return out

    def cross_attention(self, q: Var, kvt: Var, kvip: Var, *, batch, heads, n, nt, nip, d, fusion=None, vnorm=None, vnorm_coef=0.0) -> Var:
        """Dual-branch SDPA of PhotoVerseAttnProcessor2_0 (attention_processor.py:317-322, :392-420).  ``vnorm_coef``: d(loss)/d(each
        element of to_v_ip_norm) - the regulariser of train.py:512-513 enters the backward here (times the loss scale)."""
        C = heads * d
        o, _ = self.rf.cross_attention(q.t, kvt.t[:, :C], kvt.t[:, C:], kvip.t[:, :C], kvip.t[:, C:], batch=batch, heads=heads, nq=n, nt=nt, nip=nip,
                                       d=d, vnorm=vnorm, fusion=fusion)
        out = Var(o, True)

        def bwd():
            if out.g is None:
                return
            dq, dkt32, dki32 = self.rb.cross_attention_backward(q.t, kvt.t[:, :C], kvt.t[:, C:], kvip.t[:, :C], kvip.t[:, C:], out.g, batch=batch,
                                                                heads=heads, nq=n, nt=nt, nip=nip, d=d, fusion=fusion, vnorm_coef=vnorm_coef)
            self._accum(q, dq)
            self._accum(kvt, self.rb.cast_to_f16(dkt32))
            self._accum(kvip, self.rb.cast_to_f16(dki32))
        self.back.append(bwd)
        return out

    def geglu(self, h: Var) -> Var:
        out = Var(self.rf.geglu(h.t), h.needs)

        def bwd():
            if out.g is not None and h.needs:
                self._accum(h, self.rb.geglu_backward(h.t, out.g))
        self.back.append(bwd)
        return out

    def activation(self, x: Var, act) -> Var:
        out = Var(self.rf.act_forward(x.t, act), x.needs)

        def bwd():
            if out.g is not None and x.needs:
                self._accum(x, self.rb.act_backward(x.t, out.g, act))
        self.back.append(bwd)
        return out

    def dropout(self, x: Var, *, p, rng, site, copies=1) -> Var:
        """``copies`` independently masked copies of x side by side (the inputs of the LoRA branches of to_k / to_v share one tensor)."""
        out = Var(self.rf.dropout(x.t, p=p, rng=rng, site=site, copies=copies), x.needs)

        def bwd():
            if out.g is not None and x.needs:
                x.g = self.rb.dropout(out.g, p=p, rng=rng, site=site, copies=copies, backward=True, add=x.g)
        self.back.append(bwd)
        return out

    def col_affine(self, x: Var, scale, shift) -> Var:
        """Per-channel affine (an eval-mode BatchNorm that cannot be folded into a neighbouring conv)."""
        out = Var(self.rf.col_affine(x.t, scale, shift), x.needs)

        def bwd():
            if out.g is not None and x.needs:
                self._accum(x, self.rb.col_affine(out.g, scale, None))
        self.back.append(bwd)
        return out

    def prelu(self, x: Var, slope) -> Var:
        out = Var(self.rf.prelu(x.t, slope), x.needs)

        def bwd():
            if out.g is not None and x.needs:
                self._accum(x, self.rb.prelu(x.t, slope, dy=out.g))
        self.back.append(bwd)
        return out

    def maxpool2x2(self, x: Var, *, batch, h, w) -> Var:
        out = Var(self.rf.maxpool2x2(x.t, batch=batch, h=h, w=w), x.needs)

        def bwd():
            if out.g is not None and x.needs:
                g = out.g if out.g.is_contiguous() else self.rb.add_rows(out.g, torch.zeros_like(out.t))
                self._accum(x, self.rb.maxpool2x2(x.t, batch=batch, h=h, w=w, dy=g))
        self.back.append(bwd)
        return out

    def reshape(self, x: Var, shape) -> Var:
        """A contiguous buffer seen with another 2-D shape (NHWC rows -> one row per image for the ArcFace fc5)."""
        out = Var(x.t.view(shape), x.needs)

        def bwd():
            if out.g is not None and x.needs:
                self._accum(x, out.g.view(x.t.shape))
        self.back.append(bwd)
        return out

    def add_into(self, a: Var, b: Var, dst: torch.Tensor) -> Var:
        """dst (a column slice of a wider buffer) = a + b."""
        self.rf.add_rows(a.t, b.t, out=dst)
        out = Var(dst, a.needs or b.needs)

        def bwd():
            if out.g is not None:
                self._accum(a, out.g)
                self._accum(b, out.g)
        self.back.append(bwd)
        return out

    def columns(self, parts: List[Var], whole: torch.Tensor) -> Var:
        """``whole`` [M, sum C_i] already holds the parts side by side (they were written through column views): the concat of
        adapters.py:43 as a view; its gradient splits back into column views."""
        out = Var(whole, any(p.needs for p in parts))

        def bwd():
            if out.g is None:
                return
            off = 0
            for p in parts:
                c = p.t.shape[1]
                p.g = out.g[:, off:off + c] if p.g is None else self.rb.add_rows(p.g, out.g[:, off:off + c])
                off += c
        self.back.append(bwd)
        return out
