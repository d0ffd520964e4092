#!/bin/bash
# correctness + speed of both tile variants of the GEMM kernel (GPU box)
cd "$(dirname "$0")/.."
for wm in 4 2; do
python - "$wm" <<'PY' > /dev/null
import sys, photoverse_amd.build as b
b.FLAGS = b.FLAGS + [f"-DPV_FORCE_WM={sys.argv[1]}"]
b.build_lib(force=True, verbose=False)
PY
echo "== WM=$wm"
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gemm or conv" 2>&1 | tail -2
python tools/kbench.py "$1" 2>/dev/null | tail -n +2
done
python -m photoverse_amd.build --force > /dev/null
