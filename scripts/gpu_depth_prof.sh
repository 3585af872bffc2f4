#!/bin/bash
# per-kernel time of the LiDAR depth path (C3: one 64-beam sweep, 1500 features) + host-inclusive rate
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/dep.py <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from limo_amd import ba, synth_lidar
ctx = ba.Context(0)
fr = synth_lidar.make_frame(1)
for _ in range(5): ba.depth_estimate(ctx, fr)
t0 = time.perf_counter()
for _ in range(100): d = ba.depth_estimate(ctx, fr)
dt = (time.perf_counter() - t0) / 100
print("points %d features %d with-depth %d: %.3f ms per frame (host cloud in, depths out)" % (fr["cloud"].shape[0], fr["uv"].shape[0], int((d > 0).sum()), dt * 1e3))
PY
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_depth -o dep -- python /tmp/dep.py > gpurun_out/prof_depth.log 2>&1
grep "^points" gpurun_out/prof_depth.log
python scripts/prof_summary.py gpurun_out/prof_depth/dep_results.db
