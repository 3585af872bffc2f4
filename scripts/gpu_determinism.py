"""Run-to-run determinism of every device entry point: the same call repeated must return the same bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from limo_amd import ba, default_options, synth, synth_lidar
from test_gpu_landmark_init import _rays

ctx = ba.Context(0)
o = default_options()
REP = int(sys.argv[1]) if len(sys.argv) > 1 else 30

rng = np.random.default_rng(5)
off, rays, use_depth, truth = _rays(rng, 3000)
ref = None
bad = 0
for i in range(REP * 3):
    pos, ok = ctx.landmark_init(off, rays, use_depth)
    key = (pos[ok.astype(bool)].tobytes(), ok.tobytes())
    if ref is None: ref = key
    bad += key != ref
    # interleave a solve so that the stream / pools are in use between calls
    if i % 3 == 0: ctx.solve(synth.make_window(9000 + i % 7, n_lm=300), o)
print("landmark_init: %d of %d calls differ" % (bad, REP * 3), flush=True)

for name, mk in (("solve C2", lambda: synth.make_window(7001)), ("solve 5x500", lambda: synth.make_window(7002, n_lm=500))):
    ref, bad = None, 0
    for i in range(REP):
        w = mk(); r = ctx.solve(w, o)
        key = (w.kf_pose.tobytes(), w.lm_pos.tobytes(), r["final_cost"], r["iterations_total"])
        if ref is None: ref = key
        bad += key != ref
        if i % 2: ctx.landmark_init(off, rays, use_depth)
    print("%s: %d of %d calls differ" % (name, bad, REP), flush=True)

pw, prior, gt = synth.make_pose_only_case(71)
ref, bad = None, 0
for i in range(REP * 3):
    w = pw.copy(); r = ctx.adjust_pose_only(w, prior, default_options(min_landmarks_for_trimming=30))
    key = (w.kf_pose.tobytes(), r["final_cost"], r["iterations_total"])
    if ref is None: ref = key
    bad += key != ref
print("adjust_pose_only: %d of %d calls differ" % (bad, REP * 3), flush=True)

fr = synth_lidar.make_frame(1)
ref, bad = None, 0
for i in range(REP):
    d = ba.depth_estimate(ctx, fr)
    key = np.asarray(d).tobytes()
    if ref is None: ref = key
    bad += key != ref
print("depth_estimate: %d of %d calls differ" % (bad, REP), flush=True)
