#!/usr/bin/env python3
"""Profile bench.py on the GPU box: (1) rocprofv3 --kernel-trace --stats, (2) PMC passes (counters only, separate
runs as MI355X_MICROARCH.md prescribes) for HBM traffic of the dominant kernel.  Writes summaries under gpurun_out/prof_bench/
(copy the ones to keep into profiles/).  Usage: python3 tools/profile_bench.py [steps]"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "prof_bench")
steps = sys.argv[1] if len(sys.argv) > 1 else "5"
sys.path.insert(0, ROOT)
from bench import KERNEL_SOURCES, git_blob_sha1  # noqa: E402
env = dict(os.environ, TMPDIR="/tmp")
base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "2", "--no-cpu-baseline", "--no-train-forward"]


def run(args, tag):
    d = os.path.join(OUT, tag)
    r = subprocess.run(["rocprofv3", *args, "--output-format", "csv", "-d", d, "--", *base], capture_output=True, text=True, cwd="/tmp", env=env)
    if r.returncode != 0:
        print(tag, "failed", r.stderr[-400:])
    return d, r.stdout


d, out = run(["--kernel-trace", "--stats"], "stats")
line = [l for l in out.splitlines() if l.startswith("{")]
DOM = None
if line:
    open(os.path.join(OUT, "bench_under_rocprof.json"), "w").write(line[-1] + "\n")
    DOM = (json.loads(line[-1]).get("roofline") or {}).get("kernel")      # the dominant launch symbol as bench.py determined it from the launch tags
if not DOM:
    sys.exit("bench.py printed no roofline.kernel: nothing to attribute the PMC passes to")
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    os.remove(f)   # large; the stats summary is what is kept
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    os.replace(f, os.path.join(OUT, "kernel_stats.csv"))

traffic = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum TCC_MISS_sum"):
    d, _ = run(["--pmc", *ctr.split()], "pmc_" + ctr.split()[0])
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
                agg[name.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        os.remove(f)
    for k, cs in agg.items():
        for c, v in cs.items():
            traffic.setdefault(k, {})[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
dom = {k: v for k, v in traffic.items() if DOM in k}
# the kernel sources these counters were collected on: bench.py reports roofline.traffic_stale when the tree's differ
summary = {"dominant_kernel": DOM, "source_blobs": {k: git_blob_sha1(os.path.join(ROOT, k)) for k in KERNEL_SOURCES}, "per_kernel": traffic}
if dom:
    v = next(iter(dom.values()))
    fetch_kb = v.get("FETCH_SIZE", {}).get("mean_per_launch")
    write_kb = v.get("WRITE_SIZE", {}).get("mean_per_launch")
    # gfx950: FETCH_SIZE reports half of the bytes of wide (16 B/lane) streaming reads -> x2; WRITE_SIZE is exact for wide stores
    if fetch_kb is not None and write_kb is not None:
        summary["dominant_hbm_bytes_per_launch"] = (2.0 * fetch_kb + write_kb) * 1024.0
        summary["note"] = "bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section)"
json.dump(summary, open(os.path.join(OUT, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary.items() if k != "per_kernel"}, indent=1))
