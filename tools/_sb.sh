cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06sb
for B in 1 2 4; do
BENCH_ARGS="--batch $B --latent 64 --steps 40 --warmup 8" tools/ab_env_bench.sh 2 "" "PV_MERGE_LOWRES=1" 2>&1 | sed "s/^/bs=$B  /"
done > gpurun_out/r06sb/small_batch.txt
cat gpurun_out/r06sb/small_batch.txt
