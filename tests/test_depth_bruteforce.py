"""oracle/depth_oracle.cpp against the independently written brute-force assigner (tests/depth_bruteforce.py): same
accept / reject decision for every feature, depths equal to 1e-5, ground planes equal to 1e-9 although the two draw
different RANSAC samples - on the analytic scenes and on six synthetic HDL-64 sweeps.  This is the second pin of SURVEY §8
rows D1-D6 (the first: analytically known depths, tests/test_depth.py); the HIP kernels are then held to the oracle bit
for bit (tests/test_depth.py, -m gpu)."""
import numpy as np
import pytest

import depth_bruteforce as bf
from limo_amd import synth_lidar
from test_depth import expected_wall_depth, wall_frame


def agree(do, db):
    same = (do > 0) == (db > 0)
    assert same.all(), "decisions differ for features %s" % np.flatnonzero(~same)[:10]
    both = do > 0
    if both.any():
        rel = np.abs(do[both].astype(np.float64) - db[both]) / do[both]
        assert rel.max() <= 1e-5, rel.max()
    return both.mean()


@pytest.mark.parametrize("tilt", [0.0, 0.01])
def test_wall(oracle, tilt):
    fr = wall_frame(tilt=tilt)
    db = bf.estimate(fr, use_ground_labels=False)
    assert np.allclose(db, expected_wall_depth(fr), rtol=2e-5)
    assert agree(oracle.depth_estimate(fr, use_ground_labels=False), db) == 1.0


def test_sparse_and_collinear_windows(oracle):
    for fr in (wall_frame(row_px=40.0), wall_frame(step_px=50.0, row_px=50.0)):
        db = bf.estimate(fr, use_ground_labels=False)
        assert (db == -1).all() and (oracle.depth_estimate(fr, use_ground_labels=False) == -1).all()


def test_two_walls(oracle):
    near, far = wall_frame(depth=10.0), wall_frame(depth=14.0)
    fr = dict(near)
    fr["cloud"] = np.concatenate([near["cloud"], far["cloud"]])
    db = bf.estimate(fr, use_ground_labels=False)
    assert np.allclose(db[db > 0], 10.0, rtol=1e-4)
    assert agree(oracle.depth_estimate(fr, use_ground_labels=False), db) > 0.9
    # interleaved the other way round (far wall listed first): the order of the returns must not matter
    fr["cloud"] = np.concatenate([far["cloud"], near["cloud"]])
    assert agree(oracle.depth_estimate(fr, use_ground_labels=False), bf.estimate(fr, use_ground_labels=False)) > 0.9


@pytest.mark.parametrize("seed", [1, 2, 4, 5, 11, 12])
def test_synthetic_sweeps(oracle, seed):
    fr = synth_lidar.make_frame(seed)
    # ground plane: different RANSAC draws, same refined plane (the refinement takes every band return within 10.2 m)
    n_in, pl = oracle.ground_plane(fr)
    bn, bd = bf.ground_plane(fr)
    assert n_in > 1000 and np.abs(np.r_[bn, bd] - pl).max() < 1e-9
    for g in (False, True):
        share = agree(oracle.depth_estimate(fr, use_ground_labels=g), bf.estimate(fr, use_ground_labels=g))
        assert share > 0.1
