#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06k
python3 tools/diag/attn8_pad_ab.py 3 "PV_ATTN8_REDO_HINT=0" "PV_ATTN8_REDO_HINT=1,PV_ATTN8_LOOP_PAD=0" "PV_ATTN8_REDO_HINT=1,PV_ATTN8_LOOP_PAD=1" "PV_ATTN8_REDO_HINT=1,PV_ATTN8_LOOP_PAD=2" "PV_ATTN8_REDO_HINT=1,PV_ATTN8_LOOP_PAD=3" "PV_ATTN8_REDO_HINT=1,PV_ATTN8_LOOP_PAD=4" "PV_ATTN8_REDO_HINT=1,PV_ATTN8_LOOP_PAD=5" "PV_ATTN8_REDO_HINT=1,PV_ATTN8_LOOP_PAD=6" "PV_ATTN8_REDO_HINT=1,PV_ATTN8_LOOP_PAD=7" > gpurun_out/r06k/attn8_redo_hint.txt 2>&1
cat gpurun_out/r06k/attn8_redo_hint.txt
python3 tools/build_alt_lib.py /tmp/lib_hint_p6.so pv_attn.hip -DPV_ATTN8_REDO_HINT=1 -DPV_ATTN8_LOOP_PAD=6 &
python3 tools/build_alt_lib.py /tmp/lib_hint_p2.so pv_attn.hip -DPV_ATTN8_REDO_HINT=1 -DPV_ATTN8_LOOP_PAD=2 &
python3 tools/build_alt_lib.py /tmp/lib_hint_p0.so pv_attn.hip -DPV_ATTN8_REDO_HINT=1 -DPV_ATTN8_LOOP_PAD=0 &
wait
tools/ab_env_bench.sh 3 "" "PV_HIP_LIB=/tmp/lib_hint_p6.so" "PV_HIP_LIB=/tmp/lib_hint_p2.so" "PV_HIP_LIB=/tmp/lib_hint_p0.so" > gpurun_out/r06k/loop_hint.txt 2>&1
cat gpurun_out/r06k/loop_hint.txt
