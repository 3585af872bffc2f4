#!/bin/bash
# One-launch solve of windows without free landmarks (k_solve_wg): parity tests, latency of adjustPoseOnly on both paths
# (no profiler attached), where the host time of a call goes, the kernel table of the new path.
OUT=gpurun_out/wg
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ba.py -x -q -m gpu -k "pose_only" 2>&1 | tail -5
cat > /tmp/po.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from limo_amd import ba, default_options, synth
ctx = ba.Context(0)
o = default_options(min_landmarks_for_trimming=30)
pw, prior, gt = synth.make_pose_only_case(71)
for _ in range(5): ctx.adjust_pose_only(pw.copy(), prior, o)
N = int(os.environ.get("N_CALLS", "200"))
ts = []
for _ in range(N):
    p = pw.copy(); t0 = time.perf_counter(); r = ctx.adjust_pose_only(p, prior, o); ts.append(time.perf_counter() - t0)
ts.sort()
print("adjustPoseOnly [%s]: %d landmarks, %d obs: median %.3f ms, mean %.3f ms per call, %d LM iterations, %d solves" % (
    "lock-step launches" if os.environ.get("KBA_NO_WG_SOLVE") else "one launch", pw.n_lm, pw.n_obs, ts[N // 2] * 1e3, sum(ts) / N * 1e3, r["iterations_total"], r["num_solves"]))
PY
timeout 300 python /tmp/po.py
KBA_NO_WG_SOLVE=1 KBA_NO_COOP_SOLVE=1 timeout 300 python /tmp/po.py
N_CALLS=3 KBA_HOST_TRACE=1 timeout 300 python /tmp/po.py 2>&1 | tail -5
N_CALLS=3 KBA_HOST_TRACE=1 KBA_NO_WG_SOLVE=1 KBA_NO_COOP_SOLVE=1 KBA_NO_COOP_SOLVE=1 timeout 300 python /tmp/po.py 2>&1 | tail -5
N_CALLS=50 timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_po -o po -- python /tmp/po.py > $OUT/prof_po.log 2>&1
grep "^adjustPoseOnly" $OUT/prof_po.log
python scripts/prof_summary.py $OUT/prof_po/po_results.db | head -12
rm -rf $OUT/prof_po
