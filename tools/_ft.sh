cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/round
python -m pytest tests -q -m gpu --durations=25 > gpurun_out/round/final_tests.txt 2>&1
tail -3 gpurun_out/round/final_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/round/bench_default.json 2>/dev/null
tail -1 gpurun_out/round/bench_default.json | cut -c1-300
