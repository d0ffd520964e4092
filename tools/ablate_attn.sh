#!/bin/bash
cd "$(dirname "$0")/.."
for fl in "" "-DPV_ATTN_ABLATE=1" "-DPV_ATTN_ABLATE=2" "-DPV_ATTN_ABLATE=3"; do
python - "$fl" <<'PY' > /dev/null
import sys, photoverse_amd.build as b
b.FLAGS = b.FLAGS + sys.argv[1].split()
b.SOURCES = ["pv_attn.hip", "pv_gemm.hip", "pv_norm.hip", "pv_misc.hip"]
b.build_lib(force=True, verbose=False)
PY
echo "== flags: $fl"
python tools/kbench.py "attn d" 2>/dev/null | tail -n +2
done
python -m photoverse_amd.build --force > /dev/null
