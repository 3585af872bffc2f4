#!/bin/bash
# A/B of the one-launch single-window solve: ./scripts/gpu_single_ab.sh "TAG:ENV=.. ENV=.." ...   (ms per limo_ba_solve over 30 C2 windows)
mkdir -p gpurun_out
cat > /tmp/single.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
from limo_amd import ba, default_options, synth
ctx = ba.Context(0); o = default_options()
ws = [synth.make_window(3000 + i) for i in range(32)]
ctx.solve(ws[0].copy(), o); ctx.solve(ws[1].copy(), o)
ts = []; its = 0
for w in ws[2:]:
    t0 = time.perf_counter(); its += ctx.solve(w, o)["iterations_total"]; ts.append(time.perf_counter() - t0)
dt = sum(ts) / len(ts); ts.sort()
import hashlib
h = hashlib.sha256(b"".join(w.kf_pose.tobytes() for w in ws[2:])).hexdigest()[:12]
print("%-22s %.3f ms per window (median %.3f), %.1f LM iterations per window -> %.1f us per iteration, fallbacks %d, poses %s" % (
    sys.argv[1], dt * 1e3, ts[len(ts) // 2] * 1e3, its / 30, 1e6 * dt * 30 / its, ctx.coop_fallbacks(), h))
PY
for cfg in "$@"; do
  tag=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 120 python /tmp/single.py $tag || echo "$tag failed"
done
