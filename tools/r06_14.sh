#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06n
python -m pytest tests/test_fullsize_gpu.py tests/test_unet_gpu.py -q -m gpu 2>&1 | tail -2
BENCH_ARGS="--batch 4 --latent 96 --ip-tokens 6 --steps 30 --warmup 6" tools/ab_env_bench.sh 3 "" "PV_XF_ROWS=128" > gpurun_out/r06n/loop_cfg4_rows.txt 2>&1
cat gpurun_out/r06n/loop_cfg4_rows.txt
