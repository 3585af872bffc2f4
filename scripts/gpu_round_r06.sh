#!/bin/bash
# Round-6 evidence in one gpurun call (everything lands in gpurun_out/$TAG/; the kept files are copied to profiles/r06_*):
#   TESTS=1: the GPU test tier first (LIMO_FUZZ_SCALE applies);
#   1. the bench line (with its in-run counter passes);
#   2. GATE=1 (default): rocprofv3 kernel tables, one slot group, of the BASE build (limo_amd/lib/variants/liblimo_hip_base.so =
#      the previous round's kernels, scripts/build_baseline_lib.sh), of this build and of the base build again - on THIS box - and
#      scripts/kernel_gate.py over them: any kernel more than 8 % slower than the slower of the two base runs fails the call
#      (exit code 9 at the end; the other steps still run).  Without a base library only this build's table is taken;
#   PMC=1: counter passes of full 1024-window rounds;  SINGLE=1: one window per call, latency + table;  DRIVE=1: the full
#   4541-frame GPU-vs-oracle drive (scripts/gpu_oracle_drive_full.sh);  DEPTH=1: counter passes of the depth kernels.
# Counter passes carry --kernel-trace only.   usage: TAG=r06a TESTS=1 scripts/gpu_round_r06.sh
TAG=${TAG:-r06}
OUT=gpurun_out/$TAG
GATE=${GATE:-1}
GATE_MAP=${GATE_MAP:-}
mkdir -p $OUT
rc=0
if [ -n "$TESTS" ]; then
  ( time timeout 1800 python -m pytest tests -m gpu -x -q -s ${PYTEST_ARGS:-} ) > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed|error|Error|assert|^real|^fuzz seed|ulp|chol3" $OUT/pytest_gpu.log | tail -24
fi
if [ -z "$NO_BENCH" ]; then
  ( time python bench.py $BENCH_ARGS > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; tail -2 $OUT/bench.err; head -c 400 $OUT/bench.json; echo
fi
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
table() {  # table NAME [LIB]
  rm -rf $OUT/prof_$1
  env KBA_GROUPS=1 ${2:+LIMO_HIP_LIB=$2} timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$1 -o bench -- python bench.py ${TABLE_ARGS:---steps 5 --warmup 2} --no-extras --no-cpu-baseline --no-pmc > $OUT/prof_$1.log 2>&1
  python scripts/prof_summary.py $OUT/prof_$1/bench_results.db > $OUT/rocprof_kernel_stats_bench_one_group$3.txt
  rm -rf $OUT/prof_$1
}
BASE=limo_amd/lib/variants/liblimo_hip_base.so
if [ "$GATE" = "1" ] && [ -f $BASE ]; then
  table base1 $PWD/$BASE _base_a
  table new "" ""
  table base2 $PWD/$BASE _base_b
  head -22 $OUT/rocprof_kernel_stats_bench_one_group.txt
  maps=""; for m in $GATE_MAP; do maps="$maps --map $m"; done
  python scripts/kernel_gate.py --base $OUT/rocprof_kernel_stats_bench_one_group_base_a.txt $OUT/rocprof_kernel_stats_bench_one_group_base_b.txt --new $OUT/rocprof_kernel_stats_bench_one_group.txt $maps | tee $OUT/kernel_gate.txt
  [ ${PIPESTATUS[0]} -ne 0 ] && rc=9
  echo "(base = $(cat limo_amd/lib/variants/liblimo_hip_base.rev 2>/dev/null))" >> $OUT/kernel_gate.txt
else
  table new "" ""
  head -22 $OUT/rocprof_kernel_stats_bench_one_group.txt
fi
tail -1 $OUT/prof_new.log | head -c 300; echo
if [ -n "$PMC" ]; then
  timeout 600 python scripts/pmc_collect.py --passes all --out $OUT/pmc_kernels.json --timeout 500 | tee $OUT/pmc_summary.txt
fi
if [ -n "$SINGLE" ]; then
  ./scripts/gpu_single_ab.sh "single:A=1" | tee $OUT/single_window.txt
fi
if [ -n "$DRIVE" ]; then
  ./scripts/gpu_oracle_drive_full.sh 2>&1 | tee $OUT/oracle_drive.log | tail -6
  cp gpurun_out/oracle_drive_full.txt $OUT/oracle_drive_gpu.txt 2>/dev/null
fi
if [ -n "$DEPTH" ]; then  # counter passes of the depth kernels (32-frame call) -> pmc_depth_kernels.json, stamped with depth.hip's sha
  PMC=1 DEPTH_ONLY_PMC=1 ./scripts/gpu_depth_prof.sh > $OUT/depth_prof.log 2>&1; tail -8 $OUT/depth_prof.log
  cp gpurun_out/depth_pmc.json $OUT/pmc_depth_kernels.json 2>/dev/null
fi
exit $rc
