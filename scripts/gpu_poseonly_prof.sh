#!/bin/bash
# adjustPoseOnly (one new keyframe against fixed landmarks): latency per call + kernel stats
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/po.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from limo_amd import ba, default_options, synth, _ffi
from limo_amd.window import Window
ctx = ba.Context(0)
o = default_options(min_landmarks_for_trimming=30)
pw, prior, gt = synth.make_pose_only_case(71)
for _ in range(3): ctx.adjust_pose_only(pw.copy(), prior, o)
t0 = time.perf_counter()
N = 50
for _ in range(N):
    p = pw.copy(); r = ctx.adjust_pose_only(p, prior, o)
dt = (time.perf_counter() - t0) / N
print("adjustPoseOnly: %d landmarks, %d obs: %.2f ms per call, %d LM iterations, %d solves" % (pw.n_lm, pw.n_obs, dt * 1e3, r["iterations_total"], r["num_solves"]))
PY
python /tmp/po.py 2>&1 | grep "^adjustPoseOnly"   # no profiler attached: the latency that counts
KBA_NO_WG_SOLVE=1 KBA_NO_COOP_SOLVE=1 python /tmp/po.py 2>&1 | grep "^adjustPoseOnly" | sed 's/^adjustPoseOnly/adjustPoseOnly [lock-step launches, KBA_NO_WG_SOLVE=1 KBA_NO_COOP_SOLVE=1]/'
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_po -o po -- python /tmp/po.py > gpurun_out/prof_po.log 2>&1
grep "^adjustPoseOnly" gpurun_out/prof_po.log | sed 's/^adjustPoseOnly/adjustPoseOnly [under rocprofv3]/' 
python scripts/prof_summary.py gpurun_out/prof_po/po_results.db | head -16
