#!/bin/bash
# Round-3 evidence in one gpurun call: GPU tests -> bench line -> rocprofv3 kernel tables of the bench command (default and
# one slot group).  usage: scripts/gpu_round_r03.sh [tag]   (files land in gpurun_out/<tag>/, copy what is kept to profiles/)
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
  python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_full.log 2>&1
  grep -E "passed|failed|error" $OUT/pytest_gpu_full.log | tail -4 | tee $OUT/pytest_gpu.log
fi
python bench.py ${BENCH_ARGS} > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; head -c 400 $OUT/bench.json; echo
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_PROF" ]; then
  KBA_GROUPS=1 rocprofv3 --kernel-trace --stats -d $OUT/prof_bench1 -o bench -- python bench.py --no-extras --no-cpu-baseline > $OUT/prof_bench1.log 2>&1
  python scripts/prof_summary.py $OUT/prof_bench1/bench_results.db > $OUT/rocprof_kernel_stats_bench_one_group.txt; head -24 $OUT/rocprof_kernel_stats_bench_one_group.txt
  rm -rf $OUT/prof_bench1
fi
