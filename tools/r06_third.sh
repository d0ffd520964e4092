#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
python3 tools/diag/gn_apply_ab.py 2 4096,2048,1024,512,256 > gpurun_out/r06c/gn_apply_ab.txt 2>&1
cat gpurun_out/r06c/gn_apply_ab.txt
python3 tools/diag/linear128_ab.py 1 > gpurun_out/r06c/linear128_b.txt 2>&1
cat gpurun_out/r06c/linear128_b.txt
tools/ab_lib_bench.sh pv_attn.hip "-DPV_ATTN8_MAX3=0 -DPV_ATTN8_LOOP_PAD=3" "PV_GEMM_BIG128=0" 3 > gpurun_out/r06c/loop_ab_max3_big128.txt 2>&1
cat gpurun_out/r06c/loop_ab_max3_big128.txt
