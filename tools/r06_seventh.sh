#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06g
for pad in 0 4096; do
  echo "=== PV_XF_LDS_PAD=$pad" >> gpurun_out/r06g/xfused_lifetimes.txt
  PV_XF_LDS_PAD=$pad python3 tools/diag/xfused_lifetimes.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06g/xfused_lifetimes.txt
done
cat gpurun_out/r06g/xfused_lifetimes.txt
tools/ab_env_bench.sh 3 "" "PV_SERIAL=attn8_kernel" "PV_SERIAL=big_tile_kernel<true, false, 8, 3" > gpurun_out/r06g/loop_serial.txt 2>&1
cat gpurun_out/r06g/loop_serial.txt
