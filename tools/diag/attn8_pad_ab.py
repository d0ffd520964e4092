#!/usr/bin/env python3
"""attn8_kernel (d = 40 self-attention) built with different -D sets, same box, sustained: where the tile loop sits in the 32-byte instruction-fetch
windows (PV_ATTN8_LOOP_PAD), the fence between the two MFMA shapes of the score chain (PV_ATTN8_FENCE_LOOP), ...  Every build links against the in-tree
objects of the other sources (they travel with the snapshot); one child process per build and round.

usage (GPU box): python tools/diag/attn8_pad_ab.py [rounds] "PV_ATTN8_LOOP_PAD=3" "PV_ATTN8_LOOP_PAD=3,PV_ATTN8_FENCE_LOOP=0" ...
       no define sets: PV_ATTN8_LOOP_PAD = 0 .. 7"""
import concurrent.futures as cf
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from photoverse_amd import _lib
_lib.LIB = sys.argv[1]
from photoverse_amd.ops import Recorder
dev = torch.device("cuda")
B, n, d = 16, 4096, 40
C = 8 * d
torch.manual_seed(0)
qkv = torch.randn(B * n, 3 * C, device=dev).half()
rec = Recorder(dev)
rec.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=B, heads=8, nq=n, nk=n, d=d)
for _ in range(800): rec.run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(400): rec.run()
e1.record(); torch.cuda.synchronize()
print("%%.1f" %% (e0.elapsed_time(e1) / 400 * 1e3))
""" % ROOT
import photoverse_amd.build as b  # noqa: E402

args = sys.argv[1:]
rounds = int(args.pop(0)) if args and args[0].isdigit() else 3
sets = args or ["PV_ATTN8_LOOP_PAD=%d" % i for i in range(8)]
base = [os.path.join(b.LIBDIR, f.replace(".hip", ".o")) for f in b.SOURCES if f != "pv_attn.hip"]


def build(i):
    o, lib = "/tmp/a8_%d.o" % i, "/tmp/libpv_a8_%d.so" % i
    defs = ["-D" + d for d in sets[i].split(",") if d]
    subprocess.check_call([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get("pv_attn.hip", []), *defs, "-c", os.path.join(b.CSRC, "pv_attn.hip"), "-o", o], stderr=subprocess.DEVNULL)
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o, *base])
    return lib


with cf.ThreadPoolExecutor(8) as ex:
    libs = list(ex.map(build, range(len(sets))))
for r in range(rounds):
    row = []
    for lib in libs:
        out = subprocess.run([sys.executable, "-c", CHILD, lib], capture_output=True, text=True)
        row.append(out.stdout.strip() or "ERR " + out.stderr[-200:])
    print("round %d  " % r + "  ".join("[%s] %s us" % (s, v) for s, v in zip(sets, row)), flush=True)
