mkdir -p gpurun_out
python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; tail -2 gpurun_out/r02_bench.err; head -c 600 gpurun_out/r02_bench.json; echo
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench -- python bench.py --no-extras --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
grep '^{' gpurun_out/prof_bench.log > gpurun_out/r02_bench_under_rocprof.json
python scripts/prof_summary.py gpurun_out/prof_bench/bench_results.db > gpurun_out/r02_rocprof_kernel_stats_bench.txt
KBA_GROUPS=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench1 -o bench -- python bench.py --no-extras --no-cpu-baseline > gpurun_out/prof_bench1.log 2>&1
python scripts/prof_summary.py gpurun_out/prof_bench1/bench_results.db > gpurun_out/r02_rocprof_kernel_stats_bench_one_group.txt; head -12 gpurun_out/r02_rocprof_kernel_stats_bench_one_group.txt
