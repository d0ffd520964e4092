#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06e
python3 tools/build_alt_lib.py /tmp/lib_xf_span.so pv_xfused.hip -DPV_XF_SPAN=1 &
python3 tools/build_alt_lib.py /tmp/lib_xf_s8.so pv_xfused.hip -DPV_XF_S640_NQ2=8 &
python3 tools/build_alt_lib.py /tmp/lib_xf_both.so pv_xfused.hip -DPV_XF_SPAN=1 -DPV_XF_S640_NQ2=8 &
wait
PV_HIP_LIB=/tmp/lib_xf_both.so python -m pytest tests/test_hip_kernels.py -q -m gpu -k "cross_attention_fused or attention_processor" > gpurun_out/r06e/tests_span.txt 2>&1
tail -5 gpurun_out/r06e/tests_span.txt
PV_HIP_LIB=/tmp/lib_xf_both.so python -m pytest tests/test_unet_gpu.py tests/test_reference_pins_gpu.py tests/test_fullsize_gpu.py -q -m gpu -x > gpurun_out/r06e/tests_span2.txt 2>&1
tail -5 gpurun_out/r06e/tests_span2.txt
for lib in "" /tmp/lib_xf_span.so /tmp/lib_xf_s8.so; do
  echo "--- kbench attn2 branch, PV_HIP_LIB=$lib" >> gpurun_out/r06e/kbench_span.txt
  PV_HIP_LIB=$lib python3 tools/kbench.py "attn2 branch C" 2>/dev/null | grep -v "LNQ\|4 launches" >> gpurun_out/r06e/kbench_span.txt
done
cat gpurun_out/r06e/kbench_span.txt
PV_HIP_LIB=/tmp/lib_xf_span.so python3 tools/diag/xfused_occupancy.py > gpurun_out/r06e/xfused_occupancy_span.txt 2>&1
cat gpurun_out/r06e/xfused_occupancy_span.txt
tools/ab_env_bench.sh 3 "" "PV_HIP_LIB=/tmp/lib_xf_span.so" "PV_HIP_LIB=/tmp/lib_xf_s8.so" "PV_HIP_LIB=/tmp/lib_xf_both.so" > gpurun_out/r06e/loop_ab.txt 2>&1
cat gpurun_out/r06e/loop_ab.txt
