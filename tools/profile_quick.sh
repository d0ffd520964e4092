cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round gpurun_out/prof_one
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_one -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train-forward --one-stream > gpurun_out/prof_one/bench.json 2> gpurun_out/prof_one/err.txt
find gpurun_out/prof_one -name "*kernel_trace.csv" -delete
find gpurun_out/prof_one -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/round/onestream_kernel_stats.csv
python3 bench.py --no-cpu-baseline > gpurun_out/round/bench_default.json 2> gpurun_out/round/bench_default.err
tail -1 gpurun_out/round/bench_default.json | cut -c1-600
head -30 gpurun_out/round/onestream_kernel_stats.csv | cut -c1-160
