#!/usr/bin/env python3
"""3x3 conv on the 256 x 320 tile: the LDS-resident input patch (pv_convbig.hip MODE 3 / 4, default) against the gathered form (PV_CONV_PATCH=0) -
results vs fp32 conv2d and vs each other, column statistics, sustained timing on ONE box.   usage (GPU box): python tools/diag/conv_patch_ab.py [rounds]"""
import os
import sys
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from photoverse_amd.ops import ACT_SILU, Recorder  # noqa: E402

dev = torch.device("cuda")


def h16(*shape, scale=1.0, seed=0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).half()


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm()).item()


def build(B, c0, c1, cout, h, extras, patch, big_min=None):
    os.environ["PV_CONV_PATCH"] = "1" if patch else "0"
    x0 = h16(B, c0, h, h, seed=81)
    x1 = h16(B, c1, h, h, seed=82) if c1 else None
    w = h16(cout, c0 + c1, 3, 3, scale=(9 * (c0 + c1)) ** -0.5, seed=83)
    bias = torch.randn(cout, generator=torch.Generator().manual_seed(84))
    temb = torch.randn(B, cout, generator=torch.Generator().manual_seed(85))
    res = h16(B * h * h, cout, seed=86)
    rows = lambda t: t.permute(0, 2, 3, 1).reshape(B * h * h, -1).contiguous().cuda()
    wp = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().cuda()
    kw = dict(bias=bias.cuda(), conv=dict(batch=B, hin=h, win=h, hout=h, wout=h), colstats=True)
    if extras:
        kw.update(rowadd=temb.cuda(), rowadd_ld=cout, residual=res.cuda(), act=ACT_SILU)
    rec = Recorder(dev)
    if big_min:
        rec.big_min = big_min
    out = rec.gemm(rows(x0), wp, a1=rows(x1) if c1 else None, **kw)
    cs = rec.colstats.get((out.data_ptr(), B * h * h, cout))
    tag = rec.tags[-1][0]
    rec.run()
    torch.cuda.synchronize()
    xin = torch.cat([x0, x1], 1).float() if c1 else x0.float()
    ref = F.conv2d(xin, w.float(), bias, padding=1)
    if extras:
        ref = F.silu(ref + temb[:, :, None, None])
    ref = ref.permute(0, 2, 3, 1).reshape(B * h * h, cout)
    if extras:
        ref = ref + res.float()
    return rec, out, cs, ref, tag


def sustained(rec, warm=400, reps=100):
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
    shapes = [(16, 320, 0, 320, 64, True, None), (16, 640, 320, 320, 64, False, None), (16, 320, 320, 320, 64, True, None), (16, 640, 0, 640, 32, True, 128),
              (16, 1280, 640, 640, 32, False, 128), (2, 64, 0, 320, 64, True, 1), (1, 64, 64, 640, 32, False, 1)]
    recs = {}
    for sh in shapes:
        B, c0, c1, cout, h, extras, bm = sh
        rp, op, csp, ref, tagp = build(B, c0, c1, cout, h, extras, True, bm)
        rg, og, csg, _, tagg = build(B, c0, c1, cout, h, extras, False, bm)
        ep, eg = rel(op, ref), rel(og, ref)
        print(f"{sh}: patch {tagp} vs fp32 {ep:.2e}; gathered {tagg} vs fp32 {eg:.2e}; patch vs gathered {rel(op, og):.2e}; "
              f"colstats patch vs gathered {rel(csp, csg):.2e}", flush=True)
        assert ep < 1e-3 and eg < 1e-3, sh
        recs[sh] = (rp, rg)
    for r in range(rounds):
        for sh in shapes[:5]:
            B, c0, c1, cout, h, extras, bm = sh
            fl = 2.0 * B * h * h * cout * 9 * (c0 + c1)
            os.environ["PV_CONV_PATCH"] = "1"             # the switch is read at launch time
            tp = sustained(recs[sh][0])
            os.environ["PV_CONV_PATCH"] = "0"
            tg = sustained(recs[sh][1])
            print(f"round {r} {sh}: patch {tp:7.1f} us ({fl / tp / 1e6:6.0f} TFLOP/s)   gathered {tg:7.1f} us ({fl / tg / 1e6:6.0f} TFLOP/s)   {100 * (tp / tg - 1):+.1f} %", flush=True)


if __name__ == "__main__":
    main()
