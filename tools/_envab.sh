cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06r
tools/ab_env_bench.sh 3 "" "PV_GEMM_MI2=1024" "PV_GEMM_MI2=256" "PV_SIDE_BIG_MIN=256" "PV_XF_ROWS=64" "PV_GEMM_LN=1" > gpurun_out/r06r/loop_env2.txt 2>&1
cat gpurun_out/r06r/loop_env2.txt
