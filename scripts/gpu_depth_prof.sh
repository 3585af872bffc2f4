#!/bin/bash
# LiDAR depth path (C3: one 64-beam sweep, 1500 features): host-inclusive rate of single calls, of 32-frame batches from
# host buffers and from device-resident buffers; per-kernel time table; PMC=1 adds the HBM traffic counters of the kernels
# (separate passes, --kernel-trace only) -> gpurun_out/depth_pmc.json
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/dep.py <<'PY'
import sys, os, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from limo_amd import ba, synth_lidar
ctx = ba.Context(0)
F = int(os.environ.get("DEPTH_FRAMES", "32"))
reps = int(os.environ.get("DEPTH_REPS", "20"))
frames = [synth_lidar.make_frame(1 + k) for k in range(F)]
fr = frames[0]
def rate(fn, n_frames, reps):
    for _ in range(3): fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts)) / n_frames
d = ba.depth_estimate(ctx, fr)
res = {"points": int(fr["cloud"].shape[0]), "features": int(fr["uv"].shape[0]), "with_depth": int((d > 0).sum()), "batch_frames": F}
mode = os.environ.get("DEPTH_MODE", "all")
if mode in ("all", "single"): res["ms_per_frame_single_host"] = rate(lambda: ba.depth_estimate(ctx, fr), 1, 5 * reps)
if mode in ("all", "single"):
    pinned = dict(fr); pinned["cloud"] = ba.host_array(fr["cloud"].shape, np.float32); pinned["cloud"][:] = fr["cloud"]
    res["ms_per_frame_single_host_pinned_cloud"] = rate(lambda: ba.depth_estimate(ctx, pinned), 1, 5 * reps)
    assert np.array_equal(ba.depth_estimate(ctx, pinned), d)
if mode in ("all", "batch"): res["ms_per_frame_batch_host"] = rate(lambda: ba.depth_estimate_batch(ctx, frames), F, reps)
dev = []
for f in frames:
    g = dict(f)
    g["cloud"] = torch.from_numpy(np.ascontiguousarray(f["cloud"], np.float32)).cuda()
    g["uv"] = torch.from_numpy(np.ascontiguousarray(f["uv"], np.float32)).cuda()
    g["is_ground"] = torch.from_numpy(np.ascontiguousarray(f["is_ground"], np.uint8)).cuda()
    dev.append(g)
if mode in ("all", "batch"): res["ms_per_frame_batch_device"] = rate(lambda: ba.depth_estimate_batch(ctx, dev, device=True), F, reps)
if mode in ("all", "single"): res["ms_per_frame_single_device"] = rate(lambda: ba.depth_estimate_batch(ctx, dev[:1], device=True), 1, 5 * reps)
res["visible_points"] = int(synth_lidar.visible_points(fr))
print(json.dumps(res))
PY
for mode in ${DEPTH_ONLY_PMC:+none} single batch; do
  [ "$mode" = none ] && break
  DEPTH_MODE=$mode rocprofv3 --kernel-trace --stats -d gpurun_out/prof_depth -o dep_$mode -- python /tmp/dep.py > gpurun_out/prof_depth_$mode.log 2>&1
  echo "--- $mode-frame calls under rocprofv3 --kernel-trace"; python scripts/prof_summary.py gpurun_out/prof_depth/dep_${mode}_results.db | tee gpurun_out/rocprof_depth_$mode.txt | head -9
done
[ -z "${DEPTH_ONLY_PMC:-}" ] && python /tmp/dep.py 2>/dev/null | grep "^{" | tee gpurun_out/depth_rates.json
if [ "${PMC:-0}" = "1" ]; then
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    DEPTH_REPS=2 timeout 600 rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc_depth -o p$i -- python /tmp/dep.py > gpurun_out/pmc_depth_$i.log 2>&1 || echo "pass $i failed"
  done
  python - <<'PY'
import sqlite3, glob, json, re, hashlib
out = {"depth_source_sha16": hashlib.sha256(open("limo_amd/csrc/depth.hip", "rb").read()).hexdigest()[:16],
       "command": "scripts/gpu_depth_prof.sh (PMC=1): rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE --kernel-trace -- python /tmp/dep.py; per kernel the 32-frame dispatch (largest counter value)",
       "correction": "gfx950: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM section); counters in KiB", "kernels": {}}
for db_path in sorted(glob.glob("gpurun_out/pmc_depth/*_results.db")):
    db = sqlite3.connect(db_path)
    for name, ctr, val, dur in db.execute("select name, counter_name, max(counter_value), max(duration) from pmc_events group by name, counter_name"):
        n = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("void ", "")
        k = out["kernels"].setdefault(n, {})
        k[ctr] = val
        k["launch_us_under_counters"] = max(k.get("launch_us_under_counters", 0.0), dur / 1e3)
for n, k in out["kernels"].items():
    k["frames"] = 32  # the largest dispatch of each kernel is the one of the 32-frame call
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        k["hbm_MB"] = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024 / 1e6
        k["hbm_TBps_under_counters"] = k["hbm_MB"] / k["launch_us_under_counters"]
json.dump(out, open("gpurun_out/depth_pmc.json", "w"), indent=1)
for n, k in sorted(out["kernels"].items()):
    if n.startswith("k_"): print(n, {a: round(b, 3) for a, b in k.items()})
PY
fi
