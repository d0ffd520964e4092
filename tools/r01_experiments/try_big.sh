#!/bin/bash
cd "$(dirname "$0")/.."
for v in 2 3 0; do
python - "$v" <<'PY' > /dev/null
import sys, photoverse_amd.build as b
b.FLAGS = b.FLAGS + [f"-DPV_BIG_TILES={sys.argv[1]}"]
b.build_lib(force=True, verbose=False)
PY
echo "== PV_BIG_TILES=$v"
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "gemm or conv" 2>&1 | tail -1
python tools/kbench.py "$1" 2>/dev/null | tail -n +2
done
python -m photoverse_amd.build --force > /dev/null
