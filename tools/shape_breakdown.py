#!/usr/bin/env python3
"""Per-(kernel, grid) time breakdown of one-stream bench steps from a rocprofv3 kernel trace (GPU box)."""
import csv, glob, os, subprocess, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(ROOT, "gpurun_out", "prof_shapes")
subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "bench.py"),
                "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--one-stream"], capture_output=True, cwd="/tmp",
               env=dict(os.environ, TMPDIR="/tmp"))
agg = defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
        key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""))
        a = agg[key]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    os.remove(f)
steps = 6.0   # 1 warmup + 4 timed + 1 pre-capture eager step
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print(f"total kernel time per step: {tot / steps / 1e3:.2f} ms")
for (name, gx, gy), (n, us) in rows[:40]:
    print(f"{us / steps / 1e3:7.3f} ms/step  {n / steps:6.1f} calls/step  avg {us / n:8.1f} us  grid=({gx},{gy})  {name}")
