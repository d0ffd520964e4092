#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06l
python3 tools/build_alt_lib.py /tmp/lib_rc.so pv_attn.hip -DPV_ATTN8_RECOMPUTE=1
PV_HIP_LIB=/tmp/lib_rc.so python -m pytest tests/test_hip_kernels.py -q -m gpu -k "self_attention" 2>&1 | tail -3
python3 tools/diag/attn8_pad_ab.py 2 "PV_ATTN8_RECOMPUTE=0" "PV_ATTN8_RECOMPUTE=1,PV_ATTN8_LOOP_PAD=0" "PV_ATTN8_RECOMPUTE=1,PV_ATTN8_LOOP_PAD=1" "PV_ATTN8_RECOMPUTE=1,PV_ATTN8_LOOP_PAD=2" "PV_ATTN8_RECOMPUTE=1,PV_ATTN8_LOOP_PAD=3" "PV_ATTN8_RECOMPUTE=1,PV_ATTN8_LOOP_PAD=4" "PV_ATTN8_RECOMPUTE=1,PV_ATTN8_LOOP_PAD=5" "PV_ATTN8_RECOMPUTE=1,PV_ATTN8_LOOP_PAD=6" "PV_ATTN8_RECOMPUTE=1,PV_ATTN8_LOOP_PAD=7" > gpurun_out/r06l/attn8_recompute.txt 2>&1
cat gpurun_out/r06l/attn8_recompute.txt
for p in 0 4 6; do python3 tools/build_alt_lib.py /tmp/lib_rc_p$p.so pv_attn.hip -DPV_ATTN8_RECOMPUTE=1 -DPV_ATTN8_LOOP_PAD=$p & done
wait
tools/ab_env_bench.sh 3 "" "PV_HIP_LIB=/tmp/lib_rc.so" "PV_HIP_LIB=/tmp/lib_rc_p0.so" "PV_HIP_LIB=/tmp/lib_rc_p4.so" "PV_HIP_LIB=/tmp/lib_rc_p6.so" > gpurun_out/r06l/loop_recompute.txt 2>&1
cat gpurun_out/r06l/loop_recompute.txt
