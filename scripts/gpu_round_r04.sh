#!/bin/bash
# Round-4 evidence in one gpurun call (everything lands in gpurun_out/r04/, the kept files are copied to profiles/r04_*):
#   1. the bench line (with its in-run counter passes);  2. rocprofv3 kernel table of the bench command, one slot group;
#   3. counter passes (HBM + SQ) of full 1024-window rounds -> pmc_kernels.json (stamped with the kernel sources' sha);
#   4. one C2 window per limo_ba_solve call and adjustPoseOnly: latency + kernel tables;  5. the 4541-frame drive (limo_stream).
# Counter passes carry --kernel-trace only (no other trace domains).   usage: scripts/gpu_round_r04.sh [SKIP_DRIVE=1]
OUT=gpurun_out/r04
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -2 $OUT/bench.err; head -c 300 $OUT/bench.json; echo
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
KBA_GROUPS=1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_bench1 -o bench -- python bench.py --no-extras --no-cpu-baseline --no-pmc > $OUT/prof_bench1.log 2>&1
python scripts/prof_summary.py $OUT/prof_bench1/bench_results.db > $OUT/rocprof_kernel_stats_bench_one_group.txt; head -22 $OUT/rocprof_kernel_stats_bench_one_group.txt
rm -rf $OUT/prof_bench1
timeout 600 python scripts/pmc_collect.py --passes all --out $OUT/pmc_kernels.json --timeout 500 | tee $OUT/pmc_summary.txt
./scripts/gpu_single_ab.sh "single:A=1" | tee $OUT/single_window.txt
KBA_COOP_PLAIN_LAUNCH=1 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_single -o s -- python /tmp/single.py single > $OUT/prof_single.log 2>&1
python scripts/prof_summary.py $OUT/prof_single/s_results.db > $OUT/rocprof_kernel_stats_single_window.txt; head -6 $OUT/rocprof_kernel_stats_single_window.txt
rm -rf $OUT/prof_single
if [ -z "$SKIP_DRIVE" ]; then
  app=$(python -c "import sys; sys.path.insert(0,'tests'); import emu_ffi; print(emu_ffi.build_stream_app(gpu=True))")
  ( time timeout 1500 $app --frames 4541 --az 2000 --poses $OUT/limo_stream_poses.txt ) 2>&1 | grep -E "^limo_stream|real|^(fps|ate_rmse|ate_max|depth_fraction|keyframes|solves) " | tee $OUT/limo_stream_c5_gpu.log
fi
