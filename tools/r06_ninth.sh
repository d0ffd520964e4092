#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06i
python3 tools/build_alt_lib.py /tmp/lib_xf_frag2.so pv_xfused.hip -DPV_XF_FRAG2=1
PV_HIP_LIB=/tmp/lib_xf_frag2.so python -m pytest tests/test_hip_kernels.py -q -m gpu -k "cross_attention_fused" 2>&1 | tail -2
for lib in "" /tmp/lib_xf_frag2.so; do
  echo "--- kbench attn2 branch, PV_HIP_LIB=$lib" >> gpurun_out/r06i/kbench_frag2.txt
  PV_HIP_LIB=$lib python3 tools/kbench.py "attn2 branch C320" 2>/dev/null | grep -v "LNQ\|4 launches" >> gpurun_out/r06i/kbench_frag2.txt
  PV_HIP_LIB=$lib python3 tools/diag/xfused_occupancy.py >> gpurun_out/r06i/kbench_frag2.txt 2>&1
done
cat gpurun_out/r06i/kbench_frag2.txt
