#!/bin/bash
# Round-2 evidence run: PMC passes of full rounds (-> profiles/r02_pmc_kernels.json, which bench.py reads for roofline.frac),
# GPU tests at full fuzz scale, smoke, the bench line, the rocprofv3 kernel table of the SAME bench command, single-window
# and depth tables.  Everything lands in gpurun_out/ and is copied to profiles/ by hand afterwards.
mkdir -p gpurun_out profiles
./scripts/gpu_pmc_r02.sh > gpurun_out/pmc_r02_summary.txt 2>&1
cp gpurun_out/r02_pmc_kernels.json profiles/r02_pmc_kernels.json
( time LIMO_FUZZ_SCALE=1 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|real" gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; tail -2 gpurun_out/r02_bench.err; head -c 1500 gpurun_out/r02_bench.json; echo
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --no-extras --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
grep '^{' gpurun_out/prof_bench.log > gpurun_out/r02_bench_under_rocprof.json
python scripts/prof_summary.py gpurun_out/prof_bench/bench_results.db > gpurun_out/r02_rocprof_kernel_stats_bench.txt; head -22 gpurun_out/r02_rocprof_kernel_stats_bench.txt
./scripts/gpu_single_prof.sh > gpurun_out/r02_rocprof_kernel_stats_single_window.txt 2>&1; head -3 gpurun_out/r02_rocprof_kernel_stats_single_window.txt
