#!/usr/bin/env python3
"""Collect rocprofv3 PMC counters for one kbench case (separate passes, counters only) and print per-kernel means.
Usage (GPU box): python3 tools/pmc.py "<kbench filter>" [kernel-name-substring]"""
import csv
import glob
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASSES = [
    "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS",
    "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS",
    "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD",
    "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE",
    "FETCH_SIZE GRBM_GUI_ACTIVE",
    "WRITE_SIZE TCP_TCC_READ_REQ_sum",
]
if os.environ.get("PMC_PASSES"):   # e.g. PMC_PASSES=3,4,5 for the cache / traffic counters only
    PASSES = [PASSES[int(i)] for i in os.environ["PMC_PASSES"].split(",")]
flt = sys.argv[1]
ksub = sys.argv[2] if len(sys.argv) > 2 else ""
import shutil
out = os.path.join(ROOT, "gpurun_out", "pmc")
shutil.rmtree(out, ignore_errors=True)   # stale passes of an earlier call would be averaged in
agg = defaultdict(lambda: defaultdict(list))
for i, cs in enumerate(PASSES):
    d = os.path.join(out, f"p{i}")
    cmd = ["rocprofv3", "--pmc", *cs.split(), "--output-format", "csv", "-d", d, "--", sys.executable, os.path.join(ROOT, "tools", "kbench.py"), flt]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    if r.returncode != 0:
        print("pass", i, "failed:", r.stderr[-500:])
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "")
                if ksub in name:
                    agg[name.split("(")[0][:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    print("==", k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} mean {sum(v) / len(v):16.1f}  n={len(v)}")
