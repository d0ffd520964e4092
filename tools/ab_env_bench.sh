#!/bin/bash
# same-box A/B of N environments in the bench loop (alternating): tools/ab_env_bench.sh <rounds> "ENV_A" "ENV_B" ...   ("" = default)
cd "$(dirname "$0")/.."
rounds=$1; shift
for i in $(seq 1 $rounds); do
  for v in "$@"; do
    r=$(env $v python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --no-train-forward 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "round $i  [${v:-default}]  steps/s, ms/step: $r"
  done
done
