#!/usr/bin/env python3
"""Per-kernel micro-benchmark (HIP events on the launch stream) over the shapes of the bs=16, 64x64 UNet.
Usage on the GPU box: python tools/kbench.py [filter]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photoverse_amd.ops import Recorder, pack_geglu  # noqa: E402

dev = torch.device("cuda")
B = 16


def h16(*shape, scale=1.0):
    return (torch.randn(*shape, device=dev) * scale).half()


def timeit(rec, reps=20, warm_ms=100.0):
    """Sustained: ~0.1 s of back-to-back launches first (a sample taken from an idle GPU reads ~10 % slow: clock ramp; EXPERIMENTS.md round 5)."""
    rec.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rec.run()
    e1.record()
    torch.cuda.synchronize()
    if not os.environ.get("PV_KBENCH_COLD"):
        for _ in range(min(2000, max(10, int(warm_ms / max(e0.elapsed_time(e1), 1e-3))))):
            rec.run()
    e0.record()
    for _ in range(reps):
        rec.run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    return us


cases = []


def conv(name, cin, cout, hw, c1=0, stride=1, ups=0):
    def f():
        rec = Recorder(dev)
        x = h16(B * hw * hw, cin)
        x1 = h16(B * hw * hw, c1) if c1 else None
        w = h16(cout, 9 * (cin + c1), scale=0.02)
        ho = hw * 2 if ups else hw // stride
        rec.gemm(x, w, a1=x1, bias=torch.zeros(cout, device=dev), conv=dict(batch=B, hin=hw, win=hw, hout=ho, wout=ho, stride=stride, upsample=ups),
                 colstats=bool(os.environ.get("PV_KBENCH_COLSTATS")))     # the instantiation whose epilogue leaves GroupNorm statistics behind
        return rec, 2.0 * B * ho * ho * cout * 9 * (cin + c1), 0
    cases.append((name, f))


def gemm(name, M, K, N, geglu=False, res=True):
    def f():
        rec = Recorder(dev)
        x, w = h16(M, K), h16(N, K, scale=0.02)
        b = torch.zeros(N, device=dev)
        if geglu:
            w, b = pack_geglu(w, b)
        r = h16(M, N) if (res and not geglu) else None
        rec.gemm(x, w, bias=b, residual=r, geglu=geglu)
        byts = 2.0 * (M * K + N * K + M * (N // 2 if geglu else N) * (2 if r is not None else 1))
        return rec, 2.0 * M * N * K, byts
    cases.append((name, f))


def attn(name, d, n):
    def f():
        rec = Recorder(dev)
        C = 8 * d
        qkv = h16(B * n, 3 * C)
        rec.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=B, heads=8, nq=n, nk=n, d=d)
        return rec, 4.0 * B * n * n * C, 0
    cases.append((name, f))


def xattn(name, d, n, p=1):
    def f():
        rec = Recorder(dev)
        C = 8 * d
        q, kvt, kvi = h16(B * n, C), h16(B * 77, 2 * C), h16(B * p, 2 * C)
        rec.cross_attention(q, kvt[:, :C], kvt[:, C:], kvi[:, :C], kvi[:, C:], batch=B, heads=8, nq=n, nt=77, nip=p, d=d)
        return rec, 4.0 * B * n * (77 + p) * C, 2.0 * 2 * B * n * C
    cases.append((name, f))


def xfused(name, n, p=1, fused=True, d=40):
    """attn2 branch at C = 320 / 640: the fused launch, or the four launches it replaces (LayerNorm, to_q, dual SDPA, to_out + residual)."""
    def f():
        rec = Recorder(dev)
        C = 8 * d
        hs, kvt, kvi = h16(B * n, C), h16(B * 77, 2 * C), h16(B * p, 2 * C)
        wq, wo, bo = h16(C, C, scale=0.05), h16(C, C, scale=0.05), torch.zeros(C, device=dev)
        g, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        if fused:
            pk = Recorder(dev)
            kimg, vimg = pk.xattn_pack_kv(kvt[:, :C], kvt[:, C:], kvi[:, :C], kvi[:, C:], batch=B, heads=8, d=d, nt=77, nip=p)
            pk.run()
            rec.keep.append(pk)
            rec.cross_attention_fused(hs, wq, rec.pack_wo_for_fused(wo), bo, kimg, vimg, batch=B, nq=n, heads=8, d=d, nt=77, nip=p, ln_gamma=g, ln_beta=bt)
        else:
            n2 = rec.layernorm(hs, g, bt)
            q = rec.gemm(n2, wq, rows_per_image=n)
            xa, _ = rec.cross_attention(q, kvt[:, :C], kvt[:, C:], kvi[:, :C], kvi[:, C:], batch=B, heads=8, nq=n, nt=77, nip=p, d=d)
            rec.gemm(xa, wo, bias=bo, residual=hs, rows_per_image=n)
        M = B * n
        return rec, 4.0 * M * C * C + 4.0 * M * (77 + p) * C, 2.0 * 3 * M * C
    cases.append((name, f))


def xlnq(name, n, p=1, fused=True, d=160):
    """attn2 branch at C = 1280 / 640: norm2 + to_q + dual SDPA head-parallel (pv_cross_attention_lnq) + to_out, or the four launches."""
    def f():
        rec = Recorder(dev)
        C = 8 * d
        hs, kvt, kvi = h16(B * n, C), h16(B * 77, 2 * C), h16(B * p, 2 * C)
        wq, wo, bo = h16(C, C, scale=0.03), h16(C, C, scale=0.03), torch.zeros(C, device=dev)
        g, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        if fused:
            xa, _ = rec.cross_attention_lnq(hs, wq, kvt[:, :C], kvt[:, C:], kvi[:, :C], kvi[:, C:], batch=B, heads=8, nq=n, nt=77, nip=p, ln_gamma=g, ln_beta=bt)
        else:
            n2 = rec.layernorm(hs, g, bt)
            q = rec.gemm(n2, wq, rows_per_image=n)
            xa, _ = rec.cross_attention(q, kvt[:, :C], kvt[:, C:], kvi[:, :C], kvi[:, C:], batch=B, heads=8, nq=n, nt=77, nip=p, d=d)
        rec.gemm(xa, wo, bias=bo, residual=hs, rows_per_image=n)
        M = B * n
        return rec, 4.0 * M * C * C + 4.0 * M * (77 + p) * C, 2.0 * 3 * M * C
    cases.append((name, f))


def rowgemm(name, M, N, geglu=False, fused=True):
    """LayerNorm + K = 320 Linear (+ GEGLU): the row-owning launch (pv_row_gemm) or the two launches it replaces."""
    def f():
        from photoverse_amd.ops import pack_geglu_rows
        rec = Recorder(dev)
        x, w = h16(M, 320), h16(N, 320, scale=0.02)
        b = torch.zeros(N, device=dev)
        g, bt = torch.ones(320, device=dev), torch.zeros(320, device=dev)
        if fused:
            if geglu:
                w, b = pack_geglu_rows(w, b)
            rec.row_gemm(x, w, bias=b, ln_gamma=g, ln_beta=bt, geglu=geglu)
        else:
            if geglu:
                w, b = pack_geglu(w, b)
            n1 = rec.layernorm(x, g, bt)
            rec.gemm(n1, w, bias=b, geglu=geglu)
        return rec, 2.0 * M * N * 320, 2.0 * (M * 320 + N * 320 + M * (N // 2 if geglu else N))
    cases.append((name, f))


def gn(name, c, hw):
    def f():
        rec = Recorder(dev)
        x = h16(B * hw * hw, c)
        rec.groupnorm(x, torch.ones(c, device=dev), torch.zeros(c, device=dev), batch=B, hw=hw * hw, act=1)
        return rec, 0, 2.0 * 3 * B * hw * hw * c
    cases.append((name, f))


def gn_cs(name, c, hw):
    """GroupNorm whose statistics come from the producing GEMM's epilogue: times the finalize + apply launches only."""
    def f():
        rec = Recorder(dev)
        a = h16(B * hw * hw, c)
        x = rec.gemm(a, h16(c, c, scale=0.05), rows_per_image=hw * hw, colstats=True, splitk=0)
        rec.groupnorm(x, torch.ones(c, device=dev), torch.zeros(c, device=dev), batch=B, hw=hw * hw, act=1)
        rec.run()
        return rec.subset(lambda t: "groupnorm" in t[0]), 0, 2.0 * 2 * B * hw * hw * c
    cases.append((name, f))


def ln(name, c, rows):
    def f():
        rec = Recorder(dev)
        x = h16(rows, c)
        rec.layernorm(x, torch.ones(c, device=dev), torch.zeros(c, device=dev))
        return rec, 0, 2.0 * 2 * rows * c
    cases.append((name, f))


def conv_out(name, cin, cout, hw, b=B):
    def f():
        rec = Recorder(dev)
        x = h16(b * hw * hw, cin)
        w = h16(cout, 9 * cin, scale=0.02)
        rec.conv_out(x, w, torch.zeros(cout, device=dev), batch=b, cin=cin, h=hw, wd=hw, cout=cout)
        return rec, 2.0 * b * hw * hw * cout * 9 * cin, 2.0 * b * hw * hw * cin
    cases.append((name, f))



# ---- backward kernels of the training step (pv_train.hip / pv_backward.hip) ----
def attn_bwd(name, d, n, causal=False, heads=8, b=B):
    def f():
        rec = Recorder(dev)
        C = heads * d
        qkv, do = h16(b * n, 3 * C), h16(b * n, C)
        lse = torch.empty((b, heads, n), dtype=torch.float32, device=dev)
        pre = Recorder(dev)
        o = pre.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=b, heads=heads, nq=n, nk=n, d=d, causal=causal, lse=lse)
        pre.run()
        rec.attention_backward(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, do, lse, batch=b, heads=heads, nq=n, nk=n, d=d, causal=causal)
        rec.keep.append(pre)
        return rec, 10.0 * b * n * n * C * (0.5 if causal else 1.0), 0        # five N x N x d products (S and dP twice, dV, dK, dQ)
    cases.append((name, f))


def xattn_bwd(name, d, n, p=5):
    def f():
        rec = Recorder(dev)
        C = 8 * d
        q, kvt, kvi, do = h16(B * n, C), h16(B * 77, 2 * C), h16(B * p, 2 * C), h16(B * n, C)
        rec.cross_attention_backward(q, kvt[:, :C], kvt[:, C:], kvi[:, :C], kvi[:, C:], do, batch=B, heads=8, nq=n, nt=77, nip=p, d=d)
        return rec, 14.0 * B * n * (77 + p) * C, 2.0 * 3 * B * n * C         # pass 1: S, dP, dQ; pass 2: S, dP, dK, dV
    cases.append((name, f))


def gn_bwd(name, c, hw):
    def f():
        pre, rec = Recorder(dev), Recorder(dev)
        x, dy = h16(B * hw * hw, c), h16(B * hw * hw, c)
        g, bt = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        _, stats = pre.groupnorm(x, g, bt, batch=B, hw=hw * hw, act=1, return_stats=True)
        pre.run()
        rec.groupnorm_backward(x, dy, stats, g, bt, batch=B, hw=hw * hw, act=1)
        rec.keep.append(pre)
        return rec, 0, 2.0 * B * hw * hw * c * 5                              # x and dy twice (two passes) + dx
    cases.append((name, f))


def ln_bwd(name, c, rows):
    def f():
        rec = Recorder(dev)
        x, dy = h16(rows, c), h16(rows, c)
        rec.layernorm_backward(x, dy, torch.ones(c, device=dev), torch.zeros(c, device=dev), want_affine=False)
        return rec, 0, 2.0 * rows * c * 3
    cases.append((name, f))


def conv_dgrad(name, cin, cout, hw, stride=1):
    def f():
        from photoverse_amd.tape import conv3_dgrad_weight
        rec = Recorder(dev)
        ho = hw // stride
        dy = h16(B * ho * ho, cout)
        wd = conv3_dgrad_weight(torch.randn(cout, cin, 3, 3, device=dev) * 0.02)
        if stride == 2:
            dy = rec.dilate2x(dy, batch=B, h=ho, w=ho)
        rec.gemm(dy, wd, conv=dict(batch=B, hin=hw, win=hw, hout=hw, wout=hw))
        return rec, 2.0 * B * hw * hw * cin * 9 * cout, 0
    cases.append((name, f))


def wgrad(name, M, K, N):
    def f():
        rec = Recorder(dev)
        rec.wgrad(h16(M, N), h16(M, K))
        return rec, 2.0 * M * N * K, 2.0 * M * (N + K) * 2
    cases.append((name, f))


conv("conv3 320->320 @64", 320, 320, 64)
conv("conv3 640+320->320 @64 (dual)", 640, 320, 64, c1=320)
conv("conv3 320+320->320 @64 (dual)", 320, 320, 64, c1=320)
conv("conv3 320->320 @64 s2", 320, 320, 64, stride=2)
conv("conv3 640->640 @32", 640, 640, 32)
conv("conv3 320->640 @32", 320, 640, 32)
conv("conv3 640->640 @32 up", 640, 640, 32, ups=1)
conv("conv3 1280->1280 @16 up", 1280, 1280, 16, ups=1)
conv("conv3 1280->1280 @16", 1280, 1280, 16)
conv("conv3 1280+1280->1280 @16", 1280, 1280, 16, c1=1280)
conv("conv3 1280->1280 @8", 1280, 1280, 8)
conv("conv3 1280+1280->1280 @8", 1280, 1280, 8, c1=1280)
gemm("gemm 65536x320->320", 65536, 320, 320)
gemm("gemm 65536x320->960 (qkv)", 65536, 320, 960, res=False)
gemm("gemm 65536x1280->320 (ff2)", 65536, 1280, 320)
gemm("geglu 65536x320->2560", 65536, 320, 2560, geglu=True)
gemm("gemm 16384x640->640", 16384, 640, 640)
gemm("geglu 16384x640->5120", 16384, 640, 5120, geglu=True)
gemm("gemm 16384x2560->640 (ff2)", 16384, 2560, 640)
gemm("gemm 4096x1280->1280", 4096, 1280, 1280)
gemm("geglu 4096x1280->10240", 4096, 1280, 10240, geglu=True)
gemm("gemm 4096x5120->1280 (ff2)", 4096, 5120, 1280)
gemm("gemm 1232x768->640 (text kv)", 1232, 768, 640, res=False)
rowgemm("ln+qkv 65536x320->960 ROW-OWNER", 65536, 960)
rowgemm("ln+qkv 65536x320->960 2 launches", 65536, 960, fused=False)
rowgemm("ln+geglu 65536x320->2560 ROW-OWNER", 65536, 2560, geglu=True)
rowgemm("ln+geglu 65536x320->2560 2 launches", 65536, 2560, geglu=True, fused=False)
attn("attn d40 n4096", 40, 4096)
attn("attn d80 n1024", 80, 1024)
attn("attn d160 n256", 160, 256)
xattn("xattn d40 n4096", 40, 4096)
xattn("xattn d80 n1024", 80, 1024)
xfused("attn2 branch C320 n4096 FUSED", 4096)
xfused("attn2 branch C320 n4096 4 launches", 4096, fused=False)
xfused("attn2 branch C320 n4096 P5 FUSED", 4096, p=5)
xfused("attn2 branch C640 n1024 FUSED", 1024, d=80)
xfused("attn2 branch C640 n1024 4 launches", 1024, fused=False, d=80)
xfused("attn2 branch C640 n1024 P5 FUSED", 1024, p=5, d=80)
xlnq("attn2 branch C1280 n256 LNQ + to_out (2 launches)", 256)
xlnq("attn2 branch C1280 n256 4 launches", 256, fused=False)
xlnq("attn2 branch C1280 n64 LNQ + to_out (2 launches)", 64)
xlnq("attn2 branch C1280 n64 4 launches", 64, fused=False)
xlnq("attn2 branch C640 n1024 LNQ + to_out (2 launches)", 1024, d=80)
conv_out("conv_out 320->4 @64", 320, 4, 64)
conv_out("conv_out 128->3 @512 bs4 (VAE)", 128, 3, 512, b=4)
gn("gn+silu 320 @64", 320, 64)
gn("gn+silu 1280 @16", 1280, 16)
gn_cs("gn(colstats)+silu 320 @64", 320, 64)
gn_cs("gn(colstats)+silu 640 @32", 640, 32)
ln("ln 320 x65536", 320, 65536)
ln("ln 640 x16384", 640, 16384)
ln("ln 1280 x4096", 1280, 4096)
attn_bwd("attn BWD d40 n4096", 40, 4096)
attn_bwd("attn BWD d80 n1024", 80, 1024)
attn_bwd("attn BWD d160 n256", 160, 256)
attn_bwd("attn BWD d64 n77 causal (CLIP text, 12 heads)", 64, 77, causal=True, heads=12)
xattn_bwd("xattn BWD d40 n4096 P5", 40, 4096)
xattn_bwd("xattn BWD d80 n1024 P5", 80, 1024)
gn_bwd("gn+silu BWD 320 @64", 320, 64)
gn_bwd("gn+silu BWD 1280 @16", 1280, 16)
ln_bwd("ln BWD 320 x65536", 320, 65536)
conv_dgrad("conv3 dgrad 320->320 @64", 320, 320, 64)
conv_dgrad("conv3 dgrad 320->320 @64 s2 (dilate + conv)", 320, 320, 64, stride=2)
wgrad("wgrad 65536: dW[320x320] (attn2.to_q LoRA)", 65536, 320, 320)
wgrad("wgrad 4112: dW[1024x1024] (adapter Linear)", 4112, 1024, 1024)

flt = sys.argv[1] if len(sys.argv) > 1 else ""
# PV_KBENCH_COLD=N: every case is built N times on distinct buffers and the N instances run round-robin - inputs come from HBM, not from a
# 256-MB Infinity Cache that a loop over ONE set of buffers keeps warm (the in-engine situation for the large activations)
NCOLD = int(os.environ.get("PV_KBENCH_COLD", "1"))
print(f"{'case':52s} {'us':>10s} {'TFLOP/s':>9s} {'GB/s':>9s}")
for name, f in cases:
    if flt and not any(f_ in name for f_ in flt.split("|")):   # "a|b": either substring
        continue
    rec, flops, byts = f()
    if NCOLD > 1:
        recs = [rec] + [f()[0] for _ in range(NCOLD - 1)]
        for r in recs:
            r.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            for r in recs:
                r.run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (3 * NCOLD) * 1e3
        del recs
    else:
        us = timeit(rec)
    print(f"{name:52s} {us:10.1f} {flops / us / 1e6:9.1f} {byts / us / 1e3:9.1f}")
    del rec
    torch.cuda.empty_cache()
