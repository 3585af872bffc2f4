"""Landmark-sharded solve on the GPU (BASELINE.json configs[3]: 10 keyframes x 8000 landmarks over 4
shards), through the C-ABI: virtual shards on one MI355X, and the RCCL exchange path on a one-rank communicator."""
import numpy as np
import pytest

from limo_amd import ba, default_options, synth

pytestmark = pytest.mark.gpu


def test_c4_four_virtual_shards_match_unsharded_and_oracle(ctx, oracle):
    o = default_options()
    c4 = synth.config_c4()
    wu, ws, wo = c4.copy(), c4.copy(), c4.copy()
    ru = ctx.solve(wu, o)
    rs = ctx.solve_sharded(ws, o, 4)
    ro, _ = oracle.solve(wo, o, num_threads=3)
    assert rs["n_trimmed_landmarks"] == ru["n_trimmed_landmarks"] == ro["n_trimmed_landmarks"]
    assert rs["iterations_total"] == ru["iterations_total"]
    assert abs(rs["final_cost"] - ru["final_cost"]) <= 1e-9 * abs(ru["final_cost"])
    assert np.abs(ws.kf_pose - wu.kf_pose).max() <= 1e-9
    assert np.abs(ws.lm_pos - wu.lm_pos).max() <= 1e-6
    assert abs(rs["final_cost"] - ro["final_cost"]) <= 1e-4 * abs(ro["final_cost"])
    assert np.abs(ws.kf_pose[:, 4:] - wo.kf_pose[:, 4:]).max() <= 1e-4 * np.abs(wo.kf_pose[:, 4:]).max()
    assert np.array_equal(ws.kf_pose[0], c4.kf_pose[0])
    # what a shard puts into the exchange: per LM iteration ONE block before camera assembly + camera solve (two
    # pieces of it in the first iteration of a solve) and nine doubles before the step decision - counted per shard
    st = ctx.exchange_stats()
    iters, solves = rs["iterations_total"], rs["num_solves"]
    assert st["exchanges"] / 4 <= 2 * (iters + solves) + 2 * solves + 2, st
    assert st["bytes"] / 4 / iters <= 56 << 10, st  # ~42 KB block + the trimming round's residual maxima, amortised


@pytest.mark.parametrize("P", [2, 3, 8])
def test_c2_virtual_shards(ctx, P):
    o = default_options()
    w = synth.config_c2()
    wu, ws = w.copy(), w.copy()
    ru = ctx.solve(wu, o)
    rs = ctx.solve_sharded(ws, o, P)
    assert rs["n_trimmed_landmarks"] == ru["n_trimmed_landmarks"]
    # Same algorithm, other grouping of the sums (a shard folds its workgroups before the exchange): iterates agree to rounding
    # until the function tolerance (1e-6) ends one of the two solves an iteration earlier (P = 8 on this window: 67 against 68
    # iterations, costs 3e-9 apart) - the bar is the tolerance the solver itself stops at, not rounding
    assert abs(rs["final_cost"] - ru["final_cost"]) <= 1e-7 * abs(ru["final_cost"])
    assert np.abs(ws.kf_pose - wu.kf_pose).max() <= 1e-6


def test_rccl_exchange_on_one_rank_communicator():
    """The RCCL code path (ncclCommInitRank, ncclAllGather of the shards' blocks, ncclAllReduce of the trimming maxima and landmarks) with
    world = 1: four local shards on this GPU; must reproduce the virtual-shard run bit for bit."""
    o = default_options()
    w = synth.make_window(77, n_kf=6, n_lm=1500)
    wv, wr = w.copy(), w.copy()
    ctx_v = ba.Context(0)
    rv = ctx_v.solve_sharded(wv, o, 4)
    ctx_r = ba.Context(0)
    ctx_r.comm_init(ctx_r.comm_unique_id(), 0, 1)
    rr = ctx_r.solve_sharded(wr, o, 4)
    assert rr["final_cost"] == rv["final_cost"] and rr["iterations_total"] == rv["iterations_total"]
    assert np.array_equal(wr.kf_pose, wv.kf_pose)
    assert np.array_equal(wr.lm_pos, wv.lm_pos)
