#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06h
for p in 0 1 3 4 5 6 7; do python3 tools/build_alt_lib.py /tmp/lib_pad$p.so pv_attn.hip -DPV_ATTN8_LOOP_PAD=$p & done
wait
tools/ab_env_bench.sh 2 "" "PV_HIP_LIB=/tmp/lib_pad0.so" "PV_HIP_LIB=/tmp/lib_pad1.so" "PV_HIP_LIB=/tmp/lib_pad3.so" "PV_HIP_LIB=/tmp/lib_pad4.so" "PV_HIP_LIB=/tmp/lib_pad5.so" "PV_HIP_LIB=/tmp/lib_pad6.so" "PV_HIP_LIB=/tmp/lib_pad7.so" > gpurun_out/r06h/loop_pad_sweep.txt 2>&1
cat gpurun_out/r06h/loop_pad_sweep.txt
