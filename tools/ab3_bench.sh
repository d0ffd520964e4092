#!/bin/bash
# three-way A/B on one box: usage ab3.sh "ENV_A" "ENV_B" "ENV_C" [rounds]
cd "$(dirname "$0")/.."
for i in $(seq 1 ${4:-2}); do
  for v in "$1" "$2" "$3"; do
    r=$(env $v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --no-train-forward 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "round $i  [${v:-default}]  steps/s, ms/step: $r"
  done
done
