cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 tools/profile_bench.py 10 > gpurun_out/prof_bench_stdout.txt 2>&1
mkdir -p gpurun_out/prof_one
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_one -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train-forward --one-stream > gpurun_out/prof_one/bench.json 2> gpurun_out/prof_one/err.txt
find gpurun_out/prof_one -name "*kernel_trace.csv" -delete
python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -3 gpurun_out/prof_bench_stdout.txt; tail -1 gpurun_out/bench_default.json | cut -c1-300
mkdir -p gpurun_out/prof_train
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -o tr -- python3 tools/train_profile.py 3 > gpurun_out/prof_train/stdout.txt 2>&1
find gpurun_out/prof_train -name "*kernel_trace.csv" -delete
python3 tools/kbench.py > gpurun_out/kbench.txt 2>&1
