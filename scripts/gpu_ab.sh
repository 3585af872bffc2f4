#!/bin/bash
# A/B timing: ./scripts/gpu_ab.sh "TAG1:ENV=.. ENV=.." "TAG2:..."   (each config: python bench.py --steps 5 --warmup 2, one summary line)
# TESTS=1 runs the GPU tests first (fuzz at LIMO_FUZZ_SCALE, default a tenth); PROF=TAG adds a rocprofv3 kernel table of that config.
mkdir -p gpurun_out
export LIMO_FUZZ_SCALE=${LIMO_FUZZ_SCALE:-0.1}
if [ -n "${MICRO:-}" ]; then hipcc --offload-arch=gfx950 -O3 -std=c++17 -w scripts/micro/$MICRO.hip -o /tmp/micro_$MICRO && /tmp/micro_$MICRO; fi
if [ "${TESTS:-0}" = "1" ]; then ( time timeout 1500 python -m pytest ${PYTEST_ARGS:-tests} -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|error|Error|assert|real" gpurun_out/pytest_gpu.log | tail -12; fi
for cfg in "$@"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 300 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err || { echo "$tag FAILED"; tail -5 gpurun_out/ab_$tag.err; continue; }
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$tag.json"))
print("%-14s %7.0f windows/s  %6.2f ms/step | schur avg %6.1f us frac %.3f | lin avg %6.1f us | conv %d iters %.1f" % ("$tag", d["value"], d["ms_per_step"], 1e3*d["roofline_schur"]["avg_launch_ms"], d["roofline_schur"]["frac"], 1e3*d["roofline"]["avg_launch_ms"], d["config"]["converged"], d["config"]["mean_lm_iterations"]))
PY
  if [ "${PROF:-}" = "$tag" ]; then
    ( cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && env $envs timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o bench -- python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > gpurun_out/prof_$tag.log 2>&1 )
    python scripts/prof_summary.py gpurun_out/prof_$tag/bench_results.db > gpurun_out/rocprof_$tag.txt; head -16 gpurun_out/rocprof_$tag.txt
  fi
done
