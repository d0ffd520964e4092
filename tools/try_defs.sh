#!/bin/bash
# build the library with extra -D flags (space separated in $1), run a kbench filter ($2), restore the default build
cd "$(dirname "$0")/.."
python - $1 <<'PY' > /dev/null
import sys, photoverse_amd.build as b
b.FLAGS = b.FLAGS + sys.argv[1:]
b.build_lib(force=True, verbose=False)
PY
echo "== $1"
python tools/kbench.py "$2" | tail -n +2 | grep -v amdgpu.ids
python -m photoverse_amd.build --force > /dev/null
