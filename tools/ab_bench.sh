#!/bin/bash
# A/B on ONE box (box-to-box spread is ~3 %): alternate two environments over several bench runs
# usage: tools/ab_bench.sh "ENV_A" "ENV_B" [rounds]      e.g.  tools/ab_bench.sh "PV_NO_COLSTATS=1" "" 3
cd "$(dirname "$0")/.."
for i in $(seq 1 ${3:-3}); do
  for v in "$1" "$2"; do
    r=$(env $v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "round $i  [${v:-default}]  steps/s, ms/step: $r"
  done
done
