#!/bin/bash
# Round-5 evidence in one gpurun call (everything lands in gpurun_out/$TAG/; the kept files are copied to profiles/r05_*):
#   TESTS=1: the GPU test tier first;  1. the bench line (with its in-run counter passes);  2. rocprofv3 kernel table of the bench
#   command with one slot group;  PMC=1: counter passes of full 1024-window rounds;  SINGLE=1: one window per call, latency + table;  DEPTH=1: counter passes of the depth kernels.
# Counter passes carry --kernel-trace only.   usage: TAG=r05a TESTS=1 scripts/gpu_round_r05.sh
TAG=${TAG:-r05}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ -n "$TESTS" ]; then
  ( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error|Error|assert|^real" $OUT/pytest_gpu.log | tail -12
fi
( time python bench.py $BENCH_ARGS > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; tail -2 $OUT/bench.err; head -c 400 $OUT/bench.json; echo
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
KBA_GROUPS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_bench1 -o bench -- python bench.py --no-extras --no-cpu-baseline --no-pmc > $OUT/prof_bench1.log 2>&1
python scripts/prof_summary.py $OUT/prof_bench1/bench_results.db > $OUT/rocprof_kernel_stats_bench_one_group.txt; head -24 $OUT/rocprof_kernel_stats_bench_one_group.txt
tail -1 $OUT/prof_bench1.log | head -c 300; echo
rm -rf $OUT/prof_bench1
if [ -n "$PMC" ]; then
  timeout 600 python scripts/pmc_collect.py --passes all --out $OUT/pmc_kernels.json --timeout 500 | tee $OUT/pmc_summary.txt
fi
if [ -n "$SINGLE" ]; then
  ./scripts/gpu_single_ab.sh "single:A=1" | tee $OUT/single_window.txt
fi
if [ -n "$DEPTH" ]; then  # counter passes of the depth kernels (32-frame call) -> pmc_depth_kernels.json, stamped with depth.hip's sha
  PMC=1 DEPTH_ONLY_PMC=1 ./scripts/gpu_depth_prof.sh > $OUT/depth_prof.log 2>&1; tail -8 $OUT/depth_prof.log
  cp gpurun_out/depth_pmc.json $OUT/pmc_depth_kernels.json 2>/dev/null
fi
