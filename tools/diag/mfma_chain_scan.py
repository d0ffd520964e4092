#!/usr/bin/env python3
"""Static scan of the gfx950 ISA hipcc emits for csrc/*.hip: v_mfma whose srcC is the result of a v_mfma of a DIFFERENT shape issued at most DIST
instructions earlier.  Round 5 found one such chain (a 16x16x16 tail ONE v_add behind the 16x16x32 it accumulates onto, then the intermediate register
reused by an LDS read) returning wrong sums, non-deterministically, in a wave running at s_setprio 1 (EXPERIMENTS.md); the K48 kernels fence their two
MFMA groups since.  CPU only (hipcc cross-compiles).

usage: python tools/diag/mfma_chain_scan.py [DIST=1] [source.hip ...]
``scan(sources, dist)`` is what ``tests/test_host_cpu.py`` runs over the sources that hold two-shape chains (DIST 3, zero hits required)."""
import concurrent.futures as cf
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import photoverse_amd.build as b  # noqa: E402

#: the sources whose kernels chain MFMAs of two shapes (16x16x32 + a 16x16x16 tail) onto one accumulator
TWO_SHAPE_SOURCES = ("pv_attn.hip", "pv_attnbwd.hip")


def _scan_asm(src, path, dist):
    kern, recent, hits = None, [], {}
    for line in open(path):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            kern, recent = m.group(1), []
            continue
        t = line.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        mm = re.match(r"v_mfma_f32_(\S+)\s+([va]\[\d+:\d+\]), (\S+), (\S+), (\S+)", t)
        if mm:
            shape, dst, _, _, c = mm.groups()
            for d, sh, n, between in recent:
                if d == c and sh != shape and n <= dist:
                    key = (src, kern, sh, shape, n, tuple(between))
                    hits[key] = hits.get(key, 0) + 1
            recent = [(d, sh, n + 1, bt + ["mfma"]) for d, sh, n, bt in recent if n < 8]
            recent.append((dst, shape, 0, []))
        elif re.match(r"^(v_|ds_|s_nop|buffer|global)", t):
            recent = [(d, sh, n + 1, bt + [t.split()[0]]) for d, sh, n, bt in recent if n < 8]
    return hits


def _compile_and_scan(args):
    src, dist, tmp = args
    out = os.path.join(tmp, src.replace(".hip", ".s"))
    subprocess.run([b._hipcc(), *b.FLAGS, *b.EXTRA_FLAGS.get(src, []), "-S", "--cuda-device-only", os.path.join(b.CSRC, src), "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    n_mfma = sum(1 for line in open(out) if line.lstrip().startswith("v_mfma"))
    return _scan_asm(src, out, dist), n_mfma


def scan(sources=None, dist=1, workers=2):
    """-> (hits, n_mfma): hits = {(source, kernel, producer shape, consumer shape, instructions between, what they are): count},
    n_mfma = MFMA instructions seen (a scan that saw none scanned nothing)."""
    sources = list(sources) if sources else list(b.SOURCES)
    hits, total = {}, 0
    with tempfile.TemporaryDirectory() as tmp, cf.ThreadPoolExecutor(max_workers=workers) as ex:
        for h, n in ex.map(_compile_and_scan, [(s, dist, tmp) for s in sources]):
            hits.update(h)
            total += n
    return hits, total


if __name__ == "__main__":
    args = sys.argv[1:]
    DIST = int(args[0]) if args and args[0].isdigit() else 1
    srcs = [a for a in args if a.endswith(".hip")]
    hits, n_mfma = scan(srcs or None, DIST)
    for (src, k, sh, shape, n, between), v in sorted(hits.items()):
        print(f"{src}: {k[:90]}: {sh} -> {shape}, {n} instruction(s) between ({', '.join(between) or 'none'}) x{v}")
    print(f"{sum(hits.values())} mixed-shape dependent MFMA pair(s) at distance <= {DIST} ({n_mfma} MFMA instructions scanned)")
