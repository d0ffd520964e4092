#!/usr/bin/env python3
"""Static scan of the gfx950 ISA hipcc emits for csrc/*.hip: v_mfma whose srcC is the result of a v_mfma of a DIFFERENT shape issued at most DIST
instructions earlier.  Round 5 found one such chain (a 16x16x16 tail ONE v_add behind the 16x16x32 it accumulates onto, then the intermediate register
reused by an LDS read) returning wrong sums, non-deterministically, in a wave running at s_setprio 1 (EXPERIMENTS.md); the K48 kernels fence their two
MFMA groups since.  CPU only (hipcc cross-compiles).  usage: python tools/diag/mfma_chain_scan.py [DIST=1]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import photoverse_amd.build as b  # noqa: E402

DIST = int(sys.argv[1]) if len(sys.argv) > 1 else 1
total = 0
with tempfile.TemporaryDirectory() as tmp:
    for src in b.SOURCES:
        out = os.path.join(tmp, src.replace(".hip", ".s"))
        subprocess.run([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(src, []), "-S", "--cuda-device-only", os.path.join(b.CSRC, src), "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        kern, recent, hits = None, [], {}
        for line in open(out):
            m = re.match(r"^(_Z\S+):", line)
            if m:
                kern, recent = m.group(1), []
                continue
            t = line.strip()
            if not t or t.startswith(";") or t.startswith("."):
                continue
            mm = re.match(r"v_mfma_f32_(\S+)\s+(v\[\d+:\d+\]), (\S+), (\S+), (\S+)", t)
            if mm:
                shape, dst, _, _, c = mm.groups()
                for d, sh, n, between in recent:
                    if d == c and sh != shape and n <= DIST:
                        key = (kern, sh, shape, n, tuple(between)); hits[key] = hits.get(key, 0) + 1
                recent = [(d, sh, n + 1, bt + ["mfma"]) for d, sh, n, bt in recent if n < 8]
                recent.append((dst, shape, 0, []))
            elif re.match(r"^(v_|ds_|s_nop|buffer|global)", t):
                recent = [(d, sh, n + 1, bt + [t.split()[0]]) for d, sh, n, bt in recent if n < 8]
        for (k, sh, shape, n, between), v in sorted(hits.items()):
            total += v
            print(f"{src}: {k[:90]}: {sh} -> {shape}, {n} instruction(s) between ({', '.join(between) or 'none'}) x{v}")
print(f"{total} mixed-shape dependent MFMA pair(s) at distance <= {DIST}")
