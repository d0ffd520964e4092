#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06f
for pad in 0 4096; do
  echo "=== C = 320, PV_XF_LDS_PAD=$pad (0: two workgroups per CU; 4096: one)" >> gpurun_out/r06f/xfused_stamps.txt
  PV_XF_LDS_PAD=$pad python3 tools/diag/xfused_stamps.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06f/xfused_stamps.txt
done
echo "=== C = 640 (128-row form needs big_min... default 64-row form)" >> gpurun_out/r06f/xfused_stamps.txt
XF_C=640 python3 tools/diag/xfused_stamps.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06f/xfused_stamps.txt
cat gpurun_out/r06f/xfused_stamps.txt
