#!/bin/bash
# round 6, first GPU call: the whole GPU suite, the default bench line, the fence A/B + loop-pad sweep of attn8_kernel<497>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/r06a/tests.txt 2>&1
tail -25 gpurun_out/r06a/tests.txt
python3 bench.py > gpurun_out/r06a/bench_default.json 2> gpurun_out/r06a/bench_default.err
tail -1 gpurun_out/r06a/bench_default.json | cut -c1-300
python3 tools/diag/attn8_pad_ab.py 3 "PV_ATTN8_FENCE_LOOP=0" "PV_ATTN8_LOOP_PAD=0" "PV_ATTN8_LOOP_PAD=1" "PV_ATTN8_LOOP_PAD=2" "PV_ATTN8_LOOP_PAD=3" "PV_ATTN8_LOOP_PAD=4" "PV_ATTN8_LOOP_PAD=5" "PV_ATTN8_LOOP_PAD=6" "PV_ATTN8_LOOP_PAD=7" > gpurun_out/r06a/attn8_fence_pad.txt 2>&1
cat gpurun_out/r06a/attn8_fence_pad.txt
