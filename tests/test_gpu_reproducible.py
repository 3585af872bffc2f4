"""Every device entry point returns the same bits when it is called again with the same input, also when other calls
of different sizes run in between (pooled blocks get reused), and the results do not depend on what a device block
held before (KBA_POISON=1 fills every block with NaN bytes at allocation)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from limo_amd import ba, default_options, synth, synth_lidar
from test_emu_vs_oracle import make_pose_only_case
from test_gpu_landmark_init import _rays

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_repeated_calls_give_the_same_bits(ctx):
    o = default_options()
    rng = np.random.default_rng(5)
    off, rays, use_depth, _ = _rays(rng, 2000)
    pw, prior, _ = make_pose_only_case(71)
    po = default_options(min_landmarks_for_trimming=30)
    fr = synth_lidar.make_frame(1)
    ref = {}
    for rep in range(6):
        pos, ok = ctx.landmark_init(off, rays, use_depth)
        w = synth.make_window(7001, n_lm=600)
        r = ctx.solve(w, o)
        p = pw.copy()
        rp = ctx.adjust_pose_only(p, prior, po)
        d = np.asarray(ba.depth_estimate(ctx, fr))
        ctx.solve(synth.make_window(9000 + rep, n_lm=150 + 40 * rep), o)  # something of another size in between
        got = {
            "landmark_init": (pos[ok.astype(bool)].tobytes(), ok.tobytes()),
            "solve": (w.kf_pose.tobytes(), w.lm_pos.tobytes(), r["final_cost"], r["iterations_total"]),
            "pose_only": (p.kf_pose.tobytes(), rp["final_cost"], rp["iterations_total"]),
            "depth": d.tobytes(),
        }
        if not ref:
            ref = got
        for k in got:
            assert got[k] == ref[k], "%s differs on repetition %d" % (k, rep)


def test_results_do_not_depend_on_stale_device_memory():
    out = []
    for poison in (False, True):
        env = dict(os.environ)
        env.pop("KBA_POISON", None)
        if poison:
            env["KBA_POISON"] = "1"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_poison_check.py")], capture_output=True, text=True, timeout=600,
                           env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if "checksum" in l]
        assert len(lines) == 6
        out.append(lines)
    assert out[0] == out[1]
    assert all(" nan 0 " in l and " unchanged 0 " in l for l in out[0])


def test_linearisation_variants_give_the_same_bits():
    """k_lin_lm exists four times: <4, true> (default: view constants, landmark sums and tail inputs in LDS, four waves per SIMD),
    <3, true> (KBA_LIN_WAVES=3), <4, false> (KBA_LIN_VLDS=0: scalar loads of the view constants - also what batches with many views per
    window take) and <3, false> (both: sums in registers).  The same statements in the same order: every checksum of the poison script
    must be the same to the last bit whichever one runs."""
    out = []
    for extra in ({}, {"KBA_LIN_VLDS": "0"}, {"KBA_LIN_WAVES": "3"}, {"KBA_LIN_WAVES": "3", "KBA_LIN_VLDS": "0"}):
        env = dict(os.environ, **extra)
        env.pop("KBA_POISON", None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_poison_check.py"), "short"], capture_output=True, text=True, timeout=600,
                           env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if "checksum" in l]
        assert len(lines) >= 3
        out.append(lines)
    assert out[0] == out[1] == out[2] == out[3]


def test_fused_launch_train_gives_the_same_bits():
    """The streaming solve's round since round 6: per-view constants inside k_sched_fill, accepted landmarks + re-damping in ONE launch
    (k_after_step).  KBA_UNFUSED_TRAIN=1 runs k_view_consts, k_lm_damp and k_accept as launches of their own - the same device
    functions in the same order per window: the same bits."""
    out = []
    for extra in ({}, {"KBA_UNFUSED_TRAIN": "1"}, {"KBA_NO_SCHUR_PAIR": "1"}):
        env = dict(os.environ, **extra)
        env.pop("KBA_POISON", None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_poison_check.py")], capture_output=True, text=True, timeout=900,
                           env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if "checksum" in l]
        assert len(lines) == 6
        out.append(lines)
    assert out[0] == out[1] == out[2]  # (KBA_NO_SCHUR_PAIR=1: the draining rounds' two Schur lists as two launches instead of k_schur_lean_pair)
