#!/usr/bin/env python3
"""Experiment: start skew between the workgroups of the fused attn2 kernel.  All 512 workgroups of a launch start together and run the same
phases at the same time (HBM load, to_q, attention, to_out + store), so the HBM phases and the MFMA / VALU phases never overlap chip-wide.
Variants sleep a subset of the workgroups at kernel entry.  Builds private copies of the library; prints the launch time of each."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import photoverse_amd.build as b  # noqa: E402

MARK = "    half8_t xf[KK][2];\n    int mrow[2];"
VARIANTS = [("none", None, 0)] + [(f"{name} x{n}", cond, n) for n in (1, 2, 4) for name, cond in
                                   (("bid&1", "(blockIdx.x & 1)"), ("(bid>>3)&1", "((blockIdx.x >> 3) & 1)"), ("(bid>>8)&1", "((blockIdx.x >> 8) & 1)"))]
only = sys.argv[1:] and sys.argv[1]
results = []
for vi, (name, cond, n) in enumerate(VARIANTS):
    s = open(os.path.join(b.CSRC, "pv_xfused.hip")).read()
    assert MARK in s
    s = s.replace('#include "pv_common.h"', '#include "%s"' % os.path.join(b.CSRC, "pv_common.h"))
    if cond:
        s = s.replace(MARK, "    if (%s) { for (int i_ = 0; i_ < %d; ++i_) __builtin_amdgcn_s_sleep(127); }\n" % (cond, n) + MARK, 1)
    src, lib = f"/tmp/pv_xfused_skew{vi}.hip", f"/tmp/libpv_skew{vi}.so"
    open(src, "w").write(s)
    objs = []
    for f in b.SOURCES:
        o = f"/tmp/diag_{f}.o" if f != "pv_xfused.hip" else f"/tmp/skew{vi}.o"
        if f == "pv_xfused.hip" or not os.path.exists(o):
            subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(f, []), "-I", b.CSRC, "-c", src if f == "pv_xfused.hip" else os.path.join(b.CSRC, f), "-o", o],
                                  stderr=subprocess.DEVNULL)
        objs.append(o)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    code = f"""
import sys, torch
sys.path.insert(0, {ROOT!r})
from photoverse_amd import _lib
_lib.LIB = {lib!r}
from photoverse_amd.ops import Recorder
dev = torch.device("cuda")
torch.manual_seed(0)
B, n, C, d, p = 16, 4096, 320, 40, 1
h16 = lambda *s, scale=1.0: (torch.randn(*s, device=dev) * scale).half()
hs, kvt, kvi = h16(B * n, C), h16(B * 77, 2 * C), h16(B * p, 2 * C)
wq, wo, bo = h16(C, C, scale=0.05), h16(C, C, scale=0.05), torch.zeros(C, device=dev)
g, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
rec = Recorder(dev)
kimg, vimg = rec.xattn_pack_kv(kvt[:, :C], kvt[:, C:], kvi[:, :C], kvi[:, C:], batch=B, heads=8, d=d, nt=77, nip=p)
out, _ = rec.cross_attention_fused(hs, wq, rec.pack_wo_for_fused(wo), bo, kimg, vimg, batch=B, nq=n, heads=8, d=d, nt=77, nip=p, ln_gamma=g, ln_beta=bt)
rec.run(); torch.cuda.synchronize()
one = rec.subset(lambda t: "fused" in t[0])
for _ in range(10): one.run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): one.run()
e1.record(); torch.cuda.synchronize()
print("%.1f us  checksum %.4f" % (e0.elapsed_time(e1) * 20, float(out.float().abs().mean())))
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    print(f"{name:16s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
