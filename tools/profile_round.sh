cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round
python -m pytest tests -q -m gpu --durations=25 > gpurun_out/round/final_tests.txt 2>&1
tail -40 gpurun_out/round/final_tests.txt
python3 tools/profile_bench.py 10 > gpurun_out/prof_bench_stdout.txt 2>&1
mkdir -p gpurun_out/prof_one
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_one -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train-forward --one-stream > gpurun_out/prof_one/bench.json 2> gpurun_out/prof_one/err.txt
find gpurun_out/prof_one -name "*kernel_trace.csv" -delete
find gpurun_out/prof_one -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/round/onestream_kernel_stats.csv
python3 bench.py > gpurun_out/round/bench_default.json 2> gpurun_out/round/bench_default.err
tail -3 gpurun_out/prof_bench_stdout.txt; tail -1 gpurun_out/round/bench_default.json | cut -c1-400
python3 tools/kbench.py > gpurun_out/round/kbench.txt 2>&1
python3 tools/diag/convbig_seg_stamps.py 2>&1 | grep -v "^   [12] " > gpurun_out/round/convbig_stamps.txt
