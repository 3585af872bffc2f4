"""LiDAR depth assignment.  PARITY UNPINNED upstream: mono_lidar_depth is not in the reference
tree and has no tests there; the oracle (oracle/depth_oracle.cpp) restates it from the parameter file and is the target
of the GPU parity tests.  The CPU tests below pin the oracle itself on analytically known scenes."""
import numpy as np
import pytest

from limo_amd import synth_lidar
from limo_amd.synth import KITTI_CX, KITTI_CY, KITTI_F, KITTI_H, KITTI_W


def wall_frame(depth=12.0, tilt=0.0, step_px=2.0, row_px=4.0):
    """Dense synthetic returns on the plane z = depth + tilt * x (camera frame), lidar frame == camera frame."""
    us = np.arange(100, 1100, step_px)
    vs = np.arange(60, 320, row_px)
    U, V = np.meshgrid(us, vs)
    rx, ry = (U - KITTI_CX) / KITTI_F, (V - KITTI_CY) / KITTI_F
    z = depth / (1.0 - tilt * rx)  # z = depth + tilt * x, x = rx z
    pts = np.stack([rx * z, ry * z, z, np.zeros_like(z)], axis=-1).reshape(-1, 4).astype(np.float32)
    rng = np.random.default_rng(0)
    uv = np.stack([rng.uniform(150, 1050, 200), rng.uniform(80, 300, 200)], axis=1).astype(np.float32)
    return {
        "cloud": pts,
        "T_cam_lidar": np.array([1.0, 0, 0, 0, 0, 0, 0]),
        "f": KITTI_F, "cx": KITTI_CX, "cy": KITTI_CY, "w": KITTI_W, "h": KITTI_H,
        "uv": uv,
        "is_ground": np.zeros(200, np.uint8),
        "tilt": tilt, "depth": depth,
    }


def expected_wall_depth(fr):
    rx = (fr["uv"][:, 0].astype(np.float64) - KITTI_CX) / KITTI_F
    return fr["depth"] / (1.0 - fr["tilt"] * rx)


@pytest.mark.parametrize("tilt", [0.0, 0.01])
def test_oracle_recovers_plane_depth_exactly(oracle, tilt):
    fr = wall_frame(tilt=tilt)
    d = oracle.depth_estimate(fr, use_ground_labels=False)
    assert (d > 0).all()
    assert np.allclose(d, expected_wall_depth(fr), rtol=2e-5)  # float32 cloud coordinates limit the accuracy


def test_oracle_rejects_sparse_and_collinear_neighbourhoods(oracle):
    fr = wall_frame(row_px=40.0)  # one scan line per 9 px window: neighbours are collinear -> planarity gate rejects
    d = oracle.depth_estimate(fr, use_ground_labels=False)
    assert (d == -1).all()
    fr = wall_frame(step_px=50.0, row_px=50.0)  # fewer than 3 neighbours
    assert (oracle.depth_estimate(fr, use_ground_labels=False) == -1).all()


def test_oracle_foreground_segmentation_prefers_the_nearest_surface(oracle):
    """Two walls at 10 m and 14 m interleaved in the same pixel windows: the nearest significant bin wins."""
    near, far = wall_frame(depth=10.0, row_px=4.0), wall_frame(depth=14.0, row_px=4.0)
    far["cloud"][:, :3] *= 1.0  # same pixels, different depth
    fr = dict(near)
    fr["cloud"] = np.concatenate([near["cloud"], far["cloud"]])
    d = oracle.depth_estimate(fr, use_ground_labels=False)
    assert (d > 0).mean() > 0.9
    assert np.allclose(d[d > 0], 10.0, rtol=1e-4)


def test_oracle_ground_plane_and_ground_features(oracle):
    fr = synth_lidar.make_frame(5)
    n_in, pl = oracle.ground_plane(fr)
    assert n_in > 1000
    # camera y axis points down: the ground normal is ~(0,-1,0) and the camera sits ~1.65 m above it
    assert pl[1] < -0.99 and 1.5 < pl[3] < 1.8
    d = oracle.depth_estimate(fr)
    ok = d > 0
    assert ok.mean() > 0.3  # this scene: 37 % (see test_acceptance_share_of_the_synthetic_scenes)
    assert np.median(np.abs(d[ok] - fr["z_true"][ok])) < 0.1  # metres, against the ray-cast ground truth
    d2 = oracle.depth_estimate(fr)
    assert np.array_equal(d, d2)  # seeded RANSAC: deterministic


def test_acceptance_share_of_the_synthetic_scenes(oracle):
    """How many features get a depth, and why the others do not (ORACLE_DEPTH_STATS=1 prints the per-gate counts).
    The sweeps use the HDL-64E S2 beam layout (1/3 deg rings above -8.33 deg, 1/2 deg below; 2000 azimuth steps = 0.18 deg
    = 2.3 px): the 6x9 px window of the parameter file sees 2-3 returns of 1-2 rings, so the gates that reject are
    geometric - fewer than 3 returns in the window (21-39 % of the features), fewer than 3 returns left in the selected
    0.3 m histogram bin on grazing surfaces (9-31 %), view ray within 5.7 deg of the fitted plane (6-16 %, ground beyond
    ~16 m), triangle not planar enough (<3 %); the two depth-range gates reject nothing.  12-60 % per scene, 44 % over
    these six; the streaming drive (features spawned on surfaces the sweep hits) reaches 61 % (profiles/r02_limo_stream_c5_gpu.log)."""
    share = [float((oracle.depth_estimate(synth_lidar.make_frame(s)) > 0).mean()) for s in (1, 2, 4, 5, 11, 12)]
    assert min(share) > 0.3 and np.mean(share) > 0.4, share


# ------------------------------------------------------------------------------------------------------------------ GPU
def compare(dg, do):
    """Bit for bit: the same accept / reject decision and the same float for every feature (header of depth.hip)."""
    assert np.array_equal(dg.view(np.uint32), do.view(np.uint32)), "GPU and oracle differ in %d features" % (dg.view(np.uint32) != do.view(np.uint32)).sum()
    assert (do > 0).any()
    return (do > 0).mean()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_az", [(s, 2000) for s in range(1, 33)] + [(s, 4000) for s in range(1, 33)])
def test_gpu_depth_matches_oracle(ctx, oracle, seed, n_az):
    from limo_amd import ba

    fr = synth_lidar.make_frame(seed)
    if n_az != 2000:
        fr["cloud"] = synth_lidar.make_sweep(seed, n_az=n_az)
        fr["uv"], fr["is_ground"], fr["z_true"] = synth_lidar.make_features(fr["cloud"], seed)
    for use_ground in (False, True):
        dg = ba.depth_estimate(ctx, fr, use_ground_labels=use_ground)
        do = oracle.depth_estimate(fr, use_ground_labels=use_ground)
        compare(dg, do)
    n_g, pl_g = ba.depth_last_ground_plane(ctx, 0)
    n_o, pl_o = oracle.ground_plane(fr)
    assert n_g == n_o and np.array_equal(pl_g, pl_o)  # RANSAC inliers and the refined plane, bit for bit
    # repeatable (the only atomics are integer counters)
    assert np.array_equal(ba.depth_estimate(ctx, fr), ba.depth_estimate(ctx, fr))


@pytest.mark.gpu
def test_gpu_depth_on_analytic_walls(ctx, oracle):
    from limo_amd import ba

    for tilt in (0.0, 0.01):
        fr = wall_frame(tilt=tilt)
        dg = ba.depth_estimate(ctx, fr, use_ground_labels=False)
        assert (dg > 0).all() and np.allclose(dg, expected_wall_depth(fr), rtol=2e-5)
    fr = wall_frame(row_px=40.0)
    assert (ba.depth_estimate(ctx, fr, use_ground_labels=False) == -1).all()
    empty = dict(fr)
    empty["cloud"] = fr["cloud"][:0]
    assert (ba.depth_estimate(ctx, empty, use_ground_labels=False) == -1).all()


@pytest.mark.gpu
def test_gpu_depth_batch_equals_single_calls(ctx, oracle):
    """limo_depth_estimate_batch: frames of different sizes (one of them empty, one without ground labels' plane) in one
    call give the bits of separate calls, from host buffers and from device-resident buffers; a second, smaller call
    afterwards still starts from clean counters (the zone a call clears for its successor)."""
    import torch

    from limo_amd import ba

    frames = [synth_lidar.make_frame(s) for s in (11, 12, 13)]
    frames[1]["cloud"] = synth_lidar.make_sweep(12, n_az=900)                      # fewer returns than the others
    frames[1]["uv"], frames[1]["is_ground"], frames[1]["z_true"] = synth_lidar.make_features(frames[1]["cloud"], 12)
    frames[2]["uv"] = frames[2]["uv"][:700]
    frames[2]["is_ground"] = frames[2]["is_ground"][:700]
    empty = dict(frames[0])
    empty["cloud"] = frames[0]["cloud"][:0]
    frames.append(empty)
    singles = [ba.depth_estimate(ctx, fr) for fr in frames]
    for k, fr in enumerate(frames[:3]):
        compare(singles[k], oracle.depth_estimate(fr))
    assert (singles[3] == -1).all()
    batch = ba.depth_estimate_batch(ctx, frames)
    for a, b in zip(singles, batch):
        assert np.array_equal(a, b)
    dev_frames = []
    for fr in frames:
        d = dict(fr)
        d["cloud"] = torch.from_numpy(np.ascontiguousarray(fr["cloud"], np.float32)).cuda()
        d["uv"] = torch.from_numpy(np.ascontiguousarray(fr["uv"], np.float32)).cuda()
        d["is_ground"] = torch.from_numpy(np.ascontiguousarray(fr["is_ground"], np.uint8)).cuda()
        dev_frames.append(d)
    dev = ba.depth_estimate_batch(ctx, dev_frames, device=True)
    for a, b in zip(singles, dev):
        assert np.array_equal(a, b.cpu().numpy())
    # more frames than one launch group holds (32), then a single frame again
    many = [frames[k % 3] for k in range(35)]
    out = ba.depth_estimate_batch(ctx, many)
    for k, o in enumerate(out):
        assert np.array_equal(o, singles[k % 3])
    assert np.array_equal(ba.depth_estimate(ctx, frames[1]), singles[1])
    # page-locked host memory (limo_host_alloc) for the sweep: same result
    pinned = dict(frames[0])
    pinned["cloud"] = ba.host_array(frames[0]["cloud"].shape, np.float32)
    pinned["cloud"][:] = frames[0]["cloud"]
    assert np.array_equal(ba.depth_estimate(ctx, pinned), singles[0])
    assert np.array_equal(ba.depth_estimate(ctx, frames[0], use_ground_labels=False), ba.depth_estimate_batch(ctx, frames[:1], use_ground_labels=False)[0])


@pytest.mark.gpu
def test_gpu_depth_in_two_halves(ctx, oracle):
    """limo_depth_estimate_begin / _end (the stream driver's prefetch of the next frame): the bits of the one-piece call, also with a
    bundle-adjustment solve of another context running between the halves and with the sweep in page-locked memory; calls out of
    order are refused and leave the context usable."""
    from limo_amd import _ffi, ba, default_options, synth

    frames = [synth_lidar.make_frame(s) for s in (21, 22)]
    want = [ba.depth_estimate(ctx, fr) for fr in frames]
    compare(want[0], oracle.depth_estimate(frames[0]))
    other = ba.Context(0)
    for k, fr in enumerate(frames):
        if k == 1:
            pinned = dict(fr)
            pinned["cloud"] = ba.host_array(fr["cloud"].shape, np.float32)
            pinned["cloud"][:] = fr["cloud"]
            fr = pinned
        h = ba.depth_estimate_begin(ctx, fr)
        rep = other.solve(synth.make_window(5, n_kf=5, n_lm=300), default_options())
        assert rep["termination"] in (0, 1)
        assert np.array_equal(ba.depth_estimate_end(ctx, h), want[k])
    # out of order
    out = np.zeros(4, np.float32)
    assert ctx.lib.limo_depth_estimate_end(ctx.ptr, out.ctypes.data_as(_ffi.c_float_p), 4) == _ffi.LIMO_ERR_INVALID
    h = ba.depth_estimate_begin(ctx, frames[0])
    with pytest.raises(RuntimeError):
        ba.depth_estimate_begin(ctx, frames[1])
    with pytest.raises(RuntimeError):
        ba.depth_estimate(ctx, frames[1])
    assert np.array_equal(ba.depth_estimate_end(ctx, h), want[0])
    h = ba.depth_estimate_begin(ctx, frames[0])
    wrong = np.zeros(h[1] + 1, np.float32)
    assert ctx.lib.limo_depth_estimate_end(ctx.ptr, wrong.ctypes.data_as(_ffi.c_float_p), h[1] + 1) == _ffi.LIMO_ERR_INVALID
    assert np.array_equal(ba.depth_estimate(ctx, frames[1]), want[1])  # (the refused _end closed the call)


def cluttered_band_frame(seed, ground_share=0.2, n=60000):
    """A sweep whose z band is mostly clutter: the ground carries only `ground_share` of the band returns, so the adaptive
    RANSAC bound stays above 64 hypotheses and the second count launch (k_ransac<rest>) has to run."""
    rng = np.random.default_rng(seed)
    fr = synth_lidar.make_frame(seed)
    n_g = int(n * ground_share)
    xy = rng.uniform(-40, 40, (n, 2))
    z = np.where(np.arange(n) < n_g, -synth_lidar.LIDAR_HEIGHT + rng.normal(0, 0.02, n), rng.uniform(-3.4, -1.1, n))
    cloud = np.concatenate([xy, z[:, None], np.zeros((n, 1))], axis=1).astype(np.float32)
    fr["cloud"] = cloud[rng.permutation(n)]
    return fr


@pytest.mark.gpu
def test_gpu_depth_edge_cases_match_oracle(ctx, oracle):
    from limo_amd import ba

    def same(fr, params=None, **kw):
        dg = ba.depth_estimate(ctx, fr, params=params, **kw)
        do = oracle.depth_estimate(fr, params=params, **kw)
        assert np.array_equal(dg.view(np.uint32), do.view(np.uint32))
        if kw.get("use_ground_labels", True):
            n_g, pl_g = ba.depth_last_ground_plane(ctx, 0)
            n_o, pl_o = oracle.ground_plane(fr, params)
            assert n_g == n_o and np.array_equal(pl_g, pl_o)
            return n_o
        return 0

    # more than 64 RANSAC hypotheses needed (inlier share < 0.41 => bound > 64)
    fr = cluttered_band_frame(21)
    n_in = same(fr)
    band = ((fr["cloud"][:, 2] >= -3.5) & (fr["cloud"][:, 2] <= -1.0)).sum()
    assert 0.2 < n_in / band < 0.41
    # NaN / inf returns are dropped everywhere
    fr = synth_lidar.make_frame(6)
    bad = np.random.default_rng(0).choice(fr["cloud"].shape[0], 500, replace=False)
    fr["cloud"][bad[:200], 0] = np.nan
    fr["cloud"][bad[200:350], 2] = np.nan
    fr["cloud"][bad[350:], 1] = np.inf
    same(fr)
    # parameter variants: no refinement, window offsets and a larger window (more cells per feature), no histogram
    # segmentation, absolute local gate, unweighted ground patches
    fr = synth_lidar.make_frame(7)
    for changes in ({"ransac_plane_use_refinement": 0}, {"pixelarea_search_offset_x": 3, "pixelarea_search_offset_y": -2},
                    {"pixelarea_search_width": 14, "pixelarea_search_height": 20}, {"do_use_histogram_segmentation": 0},
                    {"treshold_depth_local_valuetype": 0, "treshold_depth_local_value": 0.2}, {"plane_estimator_use_mestimator": 0},
                    {"do_check_triangleplanar_condition": 0, "neighbors_count_min": 5}, {"ransac_seed": 99}):
        p = ba.depth_default_params()
        for k, v in changes.items():
            setattr(p, k, v)
        same(fr, params=p)
    # a cloud that is not one sweep of a spinning scanner (20 sweeps stacked) exceeds the cell lists: an error, not a
    # silently different answer
    fr = synth_lidar.make_frame(8)
    fr["cloud"] = np.concatenate([fr["cloud"]] * 20)
    with pytest.raises(RuntimeError, match="returns"):
        ba.depth_estimate(ctx, fr)
    assert np.array_equal(ba.depth_estimate(ctx, synth_lidar.make_frame(7)), oracle.depth_estimate(synth_lidar.make_frame(7)))  # context still usable
