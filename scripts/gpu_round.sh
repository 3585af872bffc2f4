#!/bin/bash
# GPU tests + smoke + bench + rocprofv3 kernel-trace summary of the SAME bench command
mkdir -p gpurun_out profiles
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err; cat gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py > gpurun_out/prof_bench.log 2>&1
grep '^{' gpurun_out/prof_bench.log > gpurun_out/bench_under_rocprof.json
python scripts/prof_summary.py gpurun_out/prof_bench/bench_results.db > gpurun_out/rocprof_kernel_stats.txt; cat gpurun_out/rocprof_kernel_stats.txt
