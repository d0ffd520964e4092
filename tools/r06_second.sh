#!/bin/bash
# round 6, second GPU call: the tests the first call stopped in front of + the new ones, attn8 v_pk_maximum3 A/B, 128-row Linear tile A/B, energy probe
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
python -m pytest tests/test_hip_kernels.py tests/test_fullsize_gpu.py tests/test_unet_gpu.py -q -m gpu --durations=8 > gpurun_out/r06b/tests.txt 2>&1
tail -15 gpurun_out/r06b/tests.txt
python3 tools/diag/attn8_pad_ab.py 3 "PV_ATTN8_MAX3=0" "PV_ATTN8_MAX3=1" "PV_ATTN8_MAX3=1,PV_ATTN8_LOOP_PAD=0" "PV_ATTN8_MAX3=1,PV_ATTN8_LOOP_PAD=1" "PV_ATTN8_MAX3=1,PV_ATTN8_LOOP_PAD=2" "PV_ATTN8_MAX3=1,PV_ATTN8_LOOP_PAD=4" "PV_ATTN8_MAX3=1,PV_ATTN8_LOOP_PAD=5" "PV_ATTN8_MAX3=1,PV_ATTN8_LOOP_PAD=6" "PV_ATTN8_MAX3=1,PV_ATTN8_LOOP_PAD=7" > gpurun_out/r06b/attn8_max3.txt 2>&1
cat gpurun_out/r06b/attn8_max3.txt
python3 tools/diag/linear128_ab.py 2 > gpurun_out/r06b/linear128.txt 2>&1
cat gpurun_out/r06b/linear128.txt
python3 tools/diag/energy_probe.py > gpurun_out/r06b/energy_probe.txt 2>&1
cat gpurun_out/r06b/energy_probe.txt
