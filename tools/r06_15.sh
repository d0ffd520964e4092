#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06o
PV_QUANT_SPLITK2=1 python -m pytest tests/test_fullsize_gpu.py -q -m gpu -k "cfg4" 2>&1 | tail -2
BENCH_ARGS="--batch 4 --latent 96 --ip-tokens 6 --steps 30 --warmup 6" tools/ab_env_bench.sh 3 "" "PV_QUANT_SPLITK2=1" > gpurun_out/r06o/loop_cfg4_splitk2.txt 2>&1
cat gpurun_out/r06o/loop_cfg4_splitk2.txt
tools/ab_env_bench.sh 2 "" "PV_QUANT_SPLITK2=1" > gpurun_out/r06o/loop_headline_splitk2.txt 2>&1
cat gpurun_out/r06o/loop_headline_splitk2.txt
