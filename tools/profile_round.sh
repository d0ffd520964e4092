#!/bin/bash
# end-of-round evidence on ONE box: GPU suite, rocprofv3 kernel stats (two-branch and one-stream), PMC traffic (written where bench.py reads it, so the
# default bench line that follows carries a fresh `roofline.traffic`), the default bench line, the driver-flag bench line, kbench, conv stamps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${PV_ROUND:-r06}
mkdir -p gpurun_out/round
python -m pytest tests -q -m gpu --durations=25 > gpurun_out/round/final_tests.txt 2>&1
tail -5 gpurun_out/round/final_tests.txt
python3 tools/profile_bench.py 10 > gpurun_out/prof_bench_stdout.txt 2>&1
cp gpurun_out/prof_bench/pmc_traffic.json profiles/${R}_pmc_traffic.json
cp gpurun_out/prof_bench/pmc_traffic.json gpurun_out/round/pmc_traffic.json
cp gpurun_out/prof_bench/kernel_stats.csv gpurun_out/round/default_kernel_stats.csv
cp gpurun_out/prof_bench/bench_under_rocprof.json gpurun_out/round/default_under_rocprof.json
mkdir -p gpurun_out/prof_one
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_one -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train-forward --one-stream > gpurun_out/prof_one/bench.json 2> gpurun_out/prof_one/err.txt
find gpurun_out/prof_one -name "*kernel_trace.csv" -delete
find gpurun_out/prof_one -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/round/onestream_kernel_stats.csv
tail -1 gpurun_out/prof_one/bench.json > gpurun_out/round/onestream_under_rocprof.json
python3 bench.py > gpurun_out/round/bench_default.json 2> gpurun_out/round/bench_default.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/round/bench_driver_flags.json 2> /dev/null
tail -3 gpurun_out/prof_bench_stdout.txt; tail -1 gpurun_out/round/bench_default.json | cut -c1-400
python3 tools/kbench.py > gpurun_out/round/kbench.txt 2>&1
python3 tools/diag/convbig_seg_stamps.py 2>&1 | grep -v "^   [12] " > gpurun_out/round/convbig_stamps.txt
for part in head tail merged; do python3 tools/diag/plan_breakdown.py $part 2>&1 | grep -v amdgpu.ids > gpurun_out/round/plan_breakdown_$part.txt; done
PMC_PASSES=0,1,3,4,5 python3 tools/pmc.py "attn2 branch C320 n4096 FUSED" xattn_fused > gpurun_out/round/pmc_xfused320.txt 2>&1
PMC_PASSES=0,1,3,4,5 python3 tools/pmc.py "attn d40 n4096" attn8_kernel > gpurun_out/round/pmc_attn8.txt 2>&1
PMC_PASSES=0,1,3,4,5 python3 tools/pmc.py "conv3 320->320 @64" big_tile > gpurun_out/round/pmc_conv_patch.txt 2>&1
