#!/bin/bash
# GPU tests + bench + rocprof summaries (kernel trace, then separate PMC passes for HBM traffic)
mkdir -p gpurun_out profiles
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python bench.py --steps 3 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; cat gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
python scripts/prof_summary.py gpurun_out/prof_bench/bench_results.db > gpurun_out/rocprof_kernel_stats.txt; cat gpurun_out/rocprof_kernel_stats.txt
