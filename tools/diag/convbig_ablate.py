#!/usr/bin/env python3
"""Diagnostic: timing ablations of pv_convbig.hip's 256 x 320 kernel (text-substituted COPIES, private libraries; results of the ablated
builds are WRONG by construction): full | no MFMA | no LDS-DMA in the loop | no fragment reads | A pieces only | W pieces only.
conv 320 -> 320 @ 64x64, B = 16, no column statistics."""
import os, subprocess, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import photoverse_amd.build as b  # noqa: E402
base = open(os.path.join(b.CSRC, "pv_convbig.hip")).read().replace('#include "pv_gemm_dev.h"', '#include "%s"' % os.path.join(b.CSRC, "pv_gemm_dev.h"))
MMA = "acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wb[ni], xa[mi], acc[ni][mi], 0, 0, 0);"
assert base.count(MMA) == 1
variants = {
    "full": lambda s: s,
    "no MFMA (operands kept live)": lambda s: s.replace(MMA, 'asm volatile("" :: "v"(wb[ni]), "v"(xa[mi]));'),
    "no LDS-DMA in the loop": lambda s: s.replace("        if (s + 3 < ns) issue(kn, IC<0>{}, IC<AP + BP>{});\n", ""),
    "no fragment reads in the loop": lambda s: s.replace("        read_frags(s);\n", "        if (s == 0) read_frags(s);\n        asm volatile(\"\" : \"+v\"(xa[0]), \"+v\"(wb[0]));\n"),
    "A pieces only": lambda s: s.replace("} else if (j - AP < BP - 1 || b_full) {", "} else if (false) {"),
    "W pieces only": lambda s: s.replace("            } else if (j < AP) {\n                unsigned off", "            } else if (j < AP) {\n                if (k.s > 2) continue;\n                unsigned off"),
    "no MFMA, no reads (LDS-DMA + barriers only)": lambda s: variants["no fragment reads in the loop"](variants["no MFMA (operands kept live)"](s)),
}
sel = os.environ.get("CB_VARIANTS")
from photoverse_amd import _lib  # noqa: E402
dev = torch.device("cuda")
B, hw, cin, cout = 16, 64, 320, 320
x = (torch.randn(B * hw * hw, cin, device=dev)).half()
w = (torch.randn(cout, 9 * cin, device=dev) * 0.02).half()
import ctypes
for i, (name, fn) in enumerate(variants.items()):
    if sel and str(i) not in sel.split(","):
        continue
    src = fn(base)
    assert name == "full" or src != base, name
    path, lib = f"/tmp/pv_convbig_abl{i}.hip", f"/tmp/libpv_abl{i}.so"
    open(path, "w").write(src)
    objs = []
    for f in b.SOURCES:
        o = os.path.join(b.LIBDIR, f.replace(".hip", ".o"))
        if f == "pv_convbig.hip":
            o = f"/tmp/abl{i}.o"
            subprocess.check_call([b._hipcc(), *b.FLAGS, "-I", b.CSRC, "-c", path, "-o", o], stderr=subprocess.DEVNULL)
        objs.append(o)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    # a private handle per variant: the Recorder class binds the library at import, so call the C entry point directly
    import importlib
    _lib.LIB = lib
    _lib._CACHED = None if hasattr(_lib, "_CACHED") else None
    code = f"""
import sys, torch, ctypes
sys.path.insert(0, {ROOT!r})
from photoverse_amd import _lib
_lib.LIB = {lib!r}
from photoverse_amd.ops import Recorder
dev = torch.device('cuda')
B, hw, cin, cout = 16, 64, 320, 320
x = torch.randn(B * hw * hw, cin, device=dev).half()
w = (torch.randn(cout, 9 * cin, device=dev) * 0.02).half()
rec = Recorder(dev)
rec.gemm(x, w, bias=torch.zeros(cout, device=dev), conv=dict(batch=B, hin=hw, win=hw, hout=hw, wout=hw))
assert rec.tags[-1][0].startswith('big_tile_kernel')
for _ in range(100): rec.run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): rec.run()
e1.record(); torch.cuda.synchronize()
print(f'{{e0.elapsed_time(e1) / 50 * 1e3:8.1f}} us   {name}')
"""
    subprocess.run([sys.executable, "-c", code], check=False)
