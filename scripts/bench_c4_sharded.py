#!/usr/bin/env python3
"""BASELINE.json configs[3]: one 10-keyframe / 8000-landmark window, landmarks sharded over the ranks with the RCCL
exchange of SURVEY §8e.  `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/bench_c4_sharded.py`
(N = 1: a one-rank communicator, 4 local shards).  Prints one JSON line on rank 0: solves/s of the sharded window,
the unsharded single-GPU time beside it, and the parity of the two results."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from limo_amd import ba, default_options, dist as ldist, synth

rank, local_rank, world = ldist.env_rank_world()
torch.cuda.set_device(local_rank)
dist = ldist.init()
ctx = ba.Context(local_rank)
ldist.init_shard_comm(dist, ctx)
o = default_options()
c4 = synth.config_c4()
n_shards = world if world > 1 else 4
reps = 5
ctx.solve_sharded(c4.copy(), o, n_shards)  # warm-up
if dist is not None:
    dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ws = c4.copy()
    rs = ctx.solve_sharded(ws, o, n_shards)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
dt = ldist.max_over_ranks(dist, dt, device="cuda" if dist is not None else "cpu")
if rank == 0:
    plain = ba.Context(local_rank)
    plain.solve(c4.copy(), o)
    t0 = time.perf_counter()
    wu = c4.copy()
    ru = plain.solve(wu, o)
    du = time.perf_counter() - t0
    print(json.dumps({
        "metric": "landmark-sharded window solves/sec (10 KF, 8000 landmarks)", "value": 1.0 / dt, "unit": "windows/s",
        "n_gpus": world, "n_shards": n_shards, "ms_per_solve": 1e3 * dt, "ms_per_solve_one_gpu_unsharded": 1e3 * du,
        "lm_iterations": rs["iterations_total"],
        "max_abs_pose_diff_vs_unsharded": float(np.abs(ws.kf_pose - wu.kf_pose).max()),
        "rel_cost_diff_vs_unsharded": abs(rs["final_cost"] - ru["final_cost"]) / abs(ru["final_cost"]),
    }))
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
