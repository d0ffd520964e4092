#!/usr/bin/env python3
"""GroupNorm + SiLU folded into the patch conv (pv_gemm_params.a_norm) against the two launches it replaces, sustained timing of the whole
norm -> conv sequence on ONE box: [scale/shift table + fused conv] vs [statistics finalize + GroupNorm-apply + patch conv].
usage (GPU box): python tools/diag/gn_fold_ab.py [rounds]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from photoverse_amd.ops import ACT_SILU, Recorder  # noqa: E402

dev = torch.device("cuda")


def h16(*shape, scale=1.0, seed=0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).half().cuda()


def build(B, c0, c1, cout, h):
    C = c0 + c1
    pre = Recorder(dev)
    x0 = pre.gemm(h16(B * h * h, 64, seed=1), h16(c0, 64, scale=0.2, seed=2), rows_per_image=h * h, colstats=True, splitk=0)
    x1 = pre.gemm(h16(B * h * h, 64, seed=3), h16(c1, 64, scale=0.2, seed=4), rows_per_image=h * h, colstats=True, splitk=0) if c1 else None
    pre.run()
    w = h16(cout, 9 * C, scale=(9 * C) ** -0.5, seed=5)
    gamma, beta, bias = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(cout, device=dev)
    geo = dict(batch=B, hin=h, win=h, hout=h, wout=h)
    fused, two, conv_only, conv_fused_only = Recorder(dev), Recorder(dev), Recorder(dev), Recorder(dev)
    for r in (fused, two, conv_only, conv_fused_only):
        r.colstats = pre.colstats
    tab = fused.groupnorm_table(x0, gamma, beta, batch=B, hw=h * h, x1=x1)
    fused.gemm(x0, w, a1=x1, bias=bias, conv=geo, colstats=True, a_norm=tab, a_norm_act=ACT_SILU, splitk=0)
    conv_fused_only.gemm(x0, w, a1=x1, bias=bias, conv=geo, colstats=True, a_norm=tab, a_norm_act=ACT_SILU, splitk=0)
    hn = two.groupnorm(x0, gamma, beta, batch=B, hw=h * h, x1=x1, act=ACT_SILU)
    two.gemm(hn, w, bias=bias, conv=geo, colstats=True, splitk=0)
    conv_only.gemm(hn, w, bias=bias, conv=geo, colstats=True, splitk=0)
    for r in (fused, two):
        r.keep.append(pre)
        r.run()
    torch.cuda.synchronize()
    return fused, two, conv_only, conv_fused_only


def sustained(rec, warm=300, reps=100):
    for _ in range(warm):
        rec.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        rec.run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    shapes = [(16, 320, 0, 320, 64), (16, 640, 320, 320, 64), (16, 320, 320, 320, 64)]
    recs = {sh: build(*sh) for sh in shapes}
    for r in range(rounds):
        for sh in shapes:
            f, t, c, cf = (sustained(x) for x in recs[sh])
            print(f"round {r} {sh}: table + fused conv {f:7.1f} us | finalize + apply + conv {t:7.1f} us | ({100 * (f / t - 1):+.1f} %)   conv alone: fused {cf:7.1f}  plain {c:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
