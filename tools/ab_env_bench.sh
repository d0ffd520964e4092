#!/bin/bash
# same-box A/B of N environments in the bench loop (alternating): tools/ab_env_bench.sh <rounds> "ENV_A" "ENV_B" ...   ("" = default)
# BENCH_ARGS="--batch 4 --latent 96 --ip-tokens 6 --steps 30 --warmup 6" selects another workload (default: the headline, 40 steps)
cd "$(dirname "$0")/.."
rounds=$1; shift
for i in $(seq 1 $rounds); do
  for v in "$@"; do
    r=$(env $v python bench.py ${BENCH_ARGS:---steps 40 --warmup 8} --no-cpu-baseline --no-roofline --no-train-forward 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "round $i  [${v:-default}]  steps/s, ms/step: $r"
  done
done
