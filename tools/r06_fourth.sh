#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06d
python3 tools/build_alt_lib.py /tmp/lib_attn_old.so pv_attn.hip -DPV_ATTN8_MAX3=0 -DPV_ATTN8_LOOP_PAD=3 &
python3 tools/build_alt_lib.py /tmp/lib_attn_m3p3.so pv_attn.hip -DPV_ATTN8_MAX3=1 -DPV_ATTN8_LOOP_PAD=3 &
python3 tools/build_alt_lib.py /tmp/lib_xf_noslp.so pv_xfused.hip -fno-slp-vectorize &
wait
python3 tools/diag/xfused_occupancy.py > gpurun_out/r06d/xfused_occupancy.txt 2>&1
cat gpurun_out/r06d/xfused_occupancy.txt
echo "--- fused attn2 kernels alone: in-tree, then -fno-slp-vectorize" > gpurun_out/r06d/xfused_noslp.txt
python3 tools/kbench.py "attn2 branch" >> gpurun_out/r06d/xfused_noslp.txt 2>/dev/null
PV_HIP_LIB=/tmp/lib_xf_noslp.so python3 tools/kbench.py "attn2 branch" >> gpurun_out/r06d/xfused_noslp.txt 2>/dev/null
cat gpurun_out/r06d/xfused_noslp.txt
tools/ab_env_bench.sh 3 "" "PV_GEMM_BIG128=0" "PV_HIP_LIB=/tmp/lib_attn_old.so" "PV_HIP_LIB=/tmp/lib_attn_m3p3.so" "PV_HIP_LIB=/tmp/lib_xf_noslp.so" > gpurun_out/r06d/loop_ab.txt 2>&1
cat gpurun_out/r06d/loop_ab.txt
