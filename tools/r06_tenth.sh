#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06j
python -m pytest tests/test_unet_gpu.py -q -m gpu -x -k "lowres or batch" 2>&1 | tail -2
PV_CONV_BIG_SPLITK8=1 python -m pytest tests/test_unet_gpu.py -q -m gpu -x -k "lowres or loop_matches" 2>&1 | tail -2
tools/ab_env_bench.sh 3 "" "PV_CONV_BIG_SPLITK8=1" "PV_CONV_PATCH=1" "PV_SPLITK_TARGET=1024" > gpurun_out/r06j/loop_env.txt 2>&1
cat gpurun_out/r06j/loop_env.txt
