#!/usr/bin/env python3
"""Diagnostic: start and end time (s_memrealtime, 100 MHz) and the CU of EVERY workgroup of one fused attn2 launch (C = 320, B = 16: 512 workgroups): do the two
workgroups of a CU start together, how long does a workgroup live, when does the last one end?  Builds a stamped copy of pv_xfused.hip into a private library."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import photoverse_amd.build as b  # noqa: E402

s = open(os.path.join(b.CSRC, "pv_xfused.hip")).read()
s = s.replace('#include "pv_common.h"', '#include "%s"\n__device__ unsigned long long xf_life[4096 * 4];\n' % os.path.join(b.CSRC, "pv_common.h"))
m = "    half8_t xf[KK][NQ];\n    int mrow[NQ];"
assert m in s
s = s.replace(m, "    if (threadIdx.x == 0) { xf_life[blockIdx.x * 4] = __builtin_amdgcn_s_memrealtime(); xf_life[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); "
              "xf_life[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime(); }\n" + m, 1)
idx = s.index('}  // namespace\n\nextern "C" int pv_xattn_pack_kv')
k = s.rfind("}\n", 0, idx)
s = s[:k] + "    if (threadIdx.x == 0) { xf_life[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime(); xf_life[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memtime() - xf_life[blockIdx.x * 4 + 3]; }\n" + s[k:]
s += '\nextern "C" int pv_xf_life(unsigned long long* out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(xf_life), (size_t)n * 8); }\n'
src = "/tmp/pv_xfused_life.hip"
open(src, "w").write(s)
lib = "/tmp/libpv_life.so"
objs = []
for f in b.SOURCES:
    if f == "pv_xfused.hip":
        o = "/tmp/life_pv_xfused.o"
        subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(f, []), "-I", b.CSRC, "-c", src, "-o", o], stderr=subprocess.DEVNULL)
    else:
        o = os.path.join(b.LIBDIR, f.replace(".hip", ".o"))
    objs.append(o)
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
from photoverse_amd import _lib  # noqa: E402
_lib.LIB = lib
from photoverse_amd.ops import Recorder  # noqa: E402

dev = torch.device("cuda")
C, B, n, p = 320, int(os.environ.get("XF_B", "16")), 4096, 1
d = C // 8
h16 = lambda *sh, scale=1.0: (torch.randn(*sh, device=dev) * scale).half()
hs, kvt, kvi = h16(B * n, C), h16(B * 77, 2 * C), h16(B * p, 2 * C)
wq, wo, bo = h16(C, C, scale=0.05), h16(C, C, scale=0.05), torch.zeros(C, device=dev)
g, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
rec = Recorder(dev)
kimg, vimg = rec.xattn_pack_kv(kvt[:, :C], kvt[:, C:], kvi[:, :C], kvi[:, C:], batch=B, heads=8, d=d, nt=77, nip=p)
rec.run()
torch.cuda.synchronize()
r2 = Recorder(dev)
r2.cross_attention_fused(hs, wq, r2.pack_wo_for_fused(wo), bo, kimg, vimg, batch=B, nq=n, heads=8, d=d, nt=77, nip=p, ln_gamma=g, ln_beta=bt)
for _ in range(600):       # sustained: the clock the chip holds under this load
    r2.run()
torch.cuda.synchronize()
W = B * n // 128
out = (ctypes.c_ulonglong * (W * 4))()
fn = r2.lib.pv_xf_life
fn.restype = ctypes.c_int
assert fn(out, W * 4) == 0
v = list(out)
st, en, hw, cyc = v[0::4], v[1::4], v[2::4], v[3::4]
t0 = min(st)
import statistics
life = [(e - s_) / 100.0 for s_, e in zip(st, en)]
print(f"{W} workgroups; launch span (first start -> last end) {(max(en) - t0) / 100.0:.1f} us")
print(f"start offsets (us): median {statistics.median([(x - t0) / 100.0 for x in st]):.2f}, 90 % {sorted((x - t0) / 100.0 for x in st)[int(W * 0.9)]:.2f}, max {(max(st) - t0) / 100.0:.2f}")
print(f"lifetimes (us): min {min(life):.1f} median {statistics.median(life):.1f} max {max(life):.1f};  shader cycles per workgroup: median {statistics.median(cyc):.0f} -> clock {statistics.median(cyc) / statistics.median(life) / 1e3:.2f} GHz")
late = [i for i, x in enumerate(st) if (x - t0) / 100.0 > 5.0]
print(f"workgroups that started more than 5 us after the first: {len(late)}" + (f" (first of them: block {late[0]}, started at {(st[late[0]] - t0) / 100.0:.1f} us)" if late else ""))
# CU identity: HW_ID bits: wave_id 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx9); XCC from another register - group by (se, sh, cu) only
from collections import Counter
cus = Counter((h >> 8) & 0xff for h in hw)
print(f"distinct (se, sh, cu) ids seen (per XCD the same ids repeat): {len(cus)}")
