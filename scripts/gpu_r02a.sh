#!/bin/bash
# round 2, call A: GPU tests (fuzz at a tenth), then the plain-landmark Schur kernel against the general one
mkdir -p gpurun_out
export LIMO_FUZZ_SCALE=${LIMO_FUZZ_SCALE:-0.1}
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) 2>&1 | tail -25
for v in 0 1; do
  KBA_SCHUR_PLAIN=$v timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02a_bench_plain$v.json 2> gpurun_out/r02a_bench_plain$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r02a_bench_plain$v.json"))
print("KBA_SCHUR_PLAIN=$v value %.0f windows/s ms/step %.2f | schur avg %.1f us frac %.3f | lin avg %.1f us" % (d["value"], d["ms_per_step"], 1e3*d["roofline_schur"]["avg_launch_ms"], d["roofline_schur"]["frac"], 1e3*d["roofline"]["avg_launch_ms"]))
PY
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02a -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r02a.log 2>&1
python scripts/prof_summary.py gpurun_out/prof_r02a/bench_results.db > gpurun_out/r02a_rocprof_kernel_stats.txt; cat gpurun_out/r02a_rocprof_kernel_stats.txt
