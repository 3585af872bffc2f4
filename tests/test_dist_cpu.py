"""The N > 1 path on CPU: two processes over gloo (127.0.0.1).  Windows are sharded across ranks with no data-path
collective; the gathered result must equal the single-process result bit for bit, the shards must tile the batch, and
the max-over-ranks timing reduction must work.  Compute backend here is the test-only emulation (no GPU in this tier)."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from limo_amd import default_options, synth
from limo_amd.dist import shard_range

SEEDS = [300, 301, 302, 303, 304]


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path[:0] = [os.path.dirname(here), here]
    import emu_ffi
    from limo_amd import dist as ld

    d = ld.init(backend="gloo")
    lo, hi = ld.shard_range(len(SEEDS), rank, world)
    ws = [synth.make_window(s, n_kf=3 + (s % 3), n_lm=150) for s in SEEDS[lo:hi]]
    d.barrier()
    reps = emu_ffi.solve_batch(ws, default_options())
    elapsed = 0.1 * (rank + 1)
    tmax = ld.max_over_ranks(d, elapsed)
    allres = ld.gather_objects(d, [(s, w.kf_pose.tolist(), r["final_cost"]) for s, w, r in zip(SEEDS[lo:hi], ws, reps)])
    if rank == 0:
        q.put((tmax, [x for part in allres for x in part]))
    d.barrier()
    d.destroy_process_group()


def test_shard_range_tiles_everything():
    for n in (0, 1, 5, 7, 256, 1000):
        for world in (1, 2, 3, 4, 8):
            got = []
            sizes = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                got += list(range(lo, hi))
                sizes.append(hi - lo)
            assert got == list(range(n))
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_matches_single_process(emu):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    tmax, gathered = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(tmax - 0.2) < 1e-12  # max over ranks of (0.1, 0.2)
    assert [g[0] for g in gathered] == SEEDS
    single = [synth.make_window(s, n_kf=3 + (s % 3), n_lm=150) for s in SEEDS]
    reps = emu.solve_batch(single, default_options())
    for (s, pose, cost), w, r in zip(gathered, single, reps):
        assert np.array_equal(np.array(pose), w.kf_pose)
        assert cost == r["final_cost"]
