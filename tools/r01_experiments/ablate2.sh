#!/bin/bash
# staging-volume ablations of the conv kernel (results of ablated builds are WRONG by design)
cd "$(dirname "$0")/.."
for ab in ${ABS:-0 4 5}; do
python - "$ab" <<'PY' > /dev/null
import sys, photoverse_amd.build as b
b.FLAGS = b.FLAGS + [f"-DPV_ABLATE={sys.argv[1]}"]
b.build_lib(force=True, verbose=False)
PY
echo "== ABLATE=$ab (0 full, 1 no MFMA, 2 no DMA, 3 DMA only, 4 A for 2/9 taps, 5 W every other step, 6 MFMA only, 7 LDS reads only)"
python tools/kbench.py "${CASE:-conv3}" | tail -n +2
done
python -m photoverse_amd.build --force > /dev/null
