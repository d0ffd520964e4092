#!/bin/bash
# timing ablations of the GEMM/conv kernel on the GPU box (results of ablated builds are WRONG by design)
cd "$(dirname "$0")/.."
for wm in 2 4; do for ab in 0 1 2 3; do
python - "$wm" "$ab" <<'PY' > /dev/null
import sys, photoverse_amd.build as b
b.FLAGS = b.FLAGS + [f"-DPV_FORCE_WM={sys.argv[1]}", f"-DPV_ABLATE={sys.argv[2]}"]
b.SOURCES = ["pv_gemm.hip", "pv_norm.hip", "pv_attn.hip", "pv_misc.hip"]
b.build_lib(force=True, verbose=False)
PY
echo "== WM=$wm ABLATE=$ab (0 full, 1 no-MFMA, 2 no-DMA, 3 DMA-only)"
python tools/kbench.py "$1" | tail -n +2
done; done
python -m photoverse_amd.build --force > /dev/null
