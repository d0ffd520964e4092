#!/bin/bash
# same-box A/B of two compile-time variants in the bench loop: tools/ab_build_bench.sh "<flags A>" "<flags B>" [rounds]
cd "$(dirname "$0")/.."
for i in $(seq 1 ${3:-2}); do
  for v in "$1" "$2"; do
    python - $v <<'PY' > /dev/null
import sys, photoverse_amd.build as b
b.FLAGS = b.FLAGS + sys.argv[1:]
b.build_lib(force=True, verbose=False)
PY
    r=$(python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --no-train-forward 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "round $i  [${v:-default}]  steps/s, ms/step: $r"
  done
done
python -m photoverse_amd.build --force > /dev/null
