#!/usr/bin/env python3
"""Benchmark of the keyframe-BA hot path: window solves/sec on synthetic KITTI-00-shaped windows.

A "step" = one pass of the hot path over one batch: reset the HBM-resident batch to its initial parameters and run
the full solveTrimmed schedule ({2 LM iterations -> quantile trimming -> solve to tolerance}) for every window of
the batch through the C-ABI (limo_ba_batch_reset + limo_ba_batch_solve).  Inputs are resident in HBM before the
timed region.  value = windows solved per second over all ranks.

  python bench.py --gpus N --steps K --warmup W [--batch B] [--no-cpu-baseline]
For N > 1 the driver launches one rank per GPU with torch.distributed.run; windows are independent, so ranks shard
the batch with NO data-path collective ("replicas only" for this configuration) - weak scaling.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
F64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix = fp64 vector peak: 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz (v_mfma_f64_16x16x4 = 2048 FLOP / 64 clk)
# Work unit of the Jacobian evaluation (k_linearize), SURVEY §8d: the MATERIALISED pass moves 212 B per observation
# (read 52 B; write r 16 + J_pose 96 + J_point 48) + 84 B per depth observation.  `roofline.achieved` is that unit x
# the observations linearised / kernel time, as the bench contract defines it.  The kernel itself stores the factored
# Jacobian (52 B read + 56 B written per observation), so its HBM traffic (`roofline.traffic`, PMC) is
# about half of the unit and `achieved_on_stored_bytes` is reported next to it.
BYTES_PER_REPR_OBS = 212
BYTES_PER_DEPTH_OBS = 84
STORED_BYTES_PER_OBS = 108


class _StubBatch:
    """--selftest-dist only: stands in for limo_amd.ba.Batch so that the rank plumbing can run without a GPU."""

    def __init__(self, windows, rank):
        self.windows, self.rank = windows, rank

    def reset(self):
        pass

    def solve(self, opts):
        time.sleep(0.02 * (1 + self.rank))  # ranks finish at different times: the reported time must be the slowest

    def kernel_stats(self, reset=False):
        return {"linearize_ms": 1.0, "linearize_launches": 1, "schur_ms": 1.0, "schur_launches": 1, "total_ms": 2.0}

    def download(self):
        return [{"num_linearizations": 1, "iterations_total": 1, "n_trimmed_landmarks": 0, "termination": 0} for _ in self.windows]

    def close(self):
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("LIMO_BENCH_BATCH", "1024")), help="independent windows per GPU per step")
    ap.add_argument("--n-kf", type=int, default=5)
    ap.add_argument("--n-lm", type=int, default=2000)
    ap.add_argument("--distinct", type=int, default=int(os.environ.get("LIMO_BENCH_DISTINCT", "0")), help="distinct windows generated per rank (0 = every window of the batch is different)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-windows", type=int, default=8)
    ap.add_argument("--selftest-dist", action="store_true",
                    help="CPU check of the multi-rank plumbing only (rendezvous over gloo, barriers, max-over-ranks timing, "
                         "rank-0 JSON): the solve is replaced by a stub, nothing is measured (tests/test_bench_dist.py)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.stderr.write("bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d`\n" % (args.gpus, args.gpus))
            sys.exit(2)

    import numpy as np
    import torch

    selftest = args.selftest_dist
    if not selftest and not torch.cuda.is_available():
        sys.stderr.write("bench.py: no GPU visible - the product path has no CPU fallback\n")
        sys.exit(3)
    if not selftest:
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if selftest:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from limo_amd import default_options, synth

    opts = default_options()
    if selftest:
        args.batch, args.n_lm, args.no_cpu_baseline = min(args.batch, 4), min(args.n_lm, 60), True
        ctx = None
    else:
        from limo_amd import ba

        ctx = ba.Context(local_rank)

    # ---- synthetic input: `distinct` different windows per rank, tiled to the batch size (each copy is an
    # independent solve; distinct seeds per rank)
    distinct = args.batch if args.distinct <= 0 else max(1, min(args.distinct, args.batch))
    base = [synth.make_window(1000 + 100000 * rank + i, n_kf=args.n_kf, n_lm=args.n_lm) for i in range(distinct)]
    windows = [base[i % distinct].copy() for i in range(args.batch)]
    batch = _StubBatch(windows, rank) if selftest else ba.Batch(ctx, windows)
    n_obs = sum(w.n_obs for w in windows)
    n_dep = int(sum((w.obs_d > 0).sum() for w in windows))

    def sync_all():
        if not selftest:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            if not selftest:
                torch.cuda.synchronize()

    def step():
        batch.reset()
        batch.solve(opts)

    for _ in range(args.warmup):
        step()
    batch.kernel_stats(reset=True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if selftest else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = batch.kernel_stats(reset=False)
    reps = batch.download()

    if rank == 0:
        total_windows = args.batch * world * args.steps
        value = total_windows / elapsed
        # roofline of the dominant kernel (k_linearize = Jacobian evaluation, materialised): algorithmic bytes of all
        # window linearisations performed in the timed steps / device time of the kernel over the same steps
        lin_per_step = sum(r["num_linearizations"] for r in reps)
        per_window_bytes = [BYTES_PER_REPR_OBS * w.n_obs + BYTES_PER_DEPTH_OBS * int((w.obs_d > 0).sum()) for w in windows]
        lin_obs = sum(r["num_linearizations"] * w.n_obs for r, w in zip(reps, windows)) * args.steps
        alg_bytes = sum(r["num_linearizations"] * b for r, b in zip(reps, per_window_bytes)) * args.steps
        launches = max(1, stats["linearize_launches"])
        lin_ms = stats["linearize_ms"]
        achieved = alg_bytes / (lin_ms * 1e-3) / 1e9 if lin_ms > 0 else 0.0
        # HBM traffic per launch: PMC counters cannot be collected inside this process; the committed rocprofv3
        # FETCH_SIZE / WRITE_SIZE passes (profiles/r01_pmc_linearize.json, gfx950-corrected) give the ratio
        # traffic / algorithmic bytes of the kernel, applied to this run's algorithmic bytes per launch
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_linearize.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                per_obs = json.load(f)["hbm_bytes_per_observation"]
            traffic = per_obs * lin_obs / launches
            traffic_src = "profiles/r01_pmc_linearize.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE = %.1f B per linearised observation, x this run's observations per launch" % per_obs
        # second dominant kernel: k_schur (Schur complement, f64 MFMA).  Algorithmic flops of one launch over a window
        # = n_c^2 * 3N (SYRK of the 3N x n_c landmark-eliminated block, n_c free camera slots, N landmarks
        # in the problem); one launch per LM iteration of the window.
        nf = [10 * (w.n_kf - 1) for w in windows]  # first keyframe Pose-fixed, every other keyframe 6 + 3 + 1 slots
        schur_flops = sum(r["iterations_total"] * (n * n * 3.0 * (w.n_lm - r["n_trimmed_landmarks"])) for r, n, w in zip(reps, nf, windows)) * args.steps
        schur_ms = stats["schur_ms"]
        schur_tf = schur_flops / (schur_ms * 1e-3) / 1e12 if schur_ms > 0 else 0.0
        out = {
            "metric": "keyframe-BA window solves/sec (5 KF, ~2k landmarks)",
            "value": value,
            "unit": "windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "selftest of the multi-rank plumbing - stub solve, NOT a measurement" if selftest else "synthetic",
            "config": {
                "workload": "C2: KITTI-00-shaped windows, %d keyframes x %d landmarks, LiDAR depth + ground plane, solveTrimmed schedule {2, trim 5%%, <=100 LM iterations}" % (args.n_kf, args.n_lm),
                "batch_windows_per_gpu": args.batch,
                "distinct_windows_per_gpu": distinct,
                "observations_per_batch": n_obs,
                "depth_observations_per_batch": n_dep,
                "parallelism": "replicas" if world > 1 else "single",
                "mean_lm_iterations": float(np.mean([r["iterations_total"] for r in reps])),
                "max_lm_iterations": int(max(r["iterations_total"] for r in reps)),
                "converged": int(sum(r["termination"] == 0 for r in reps)),
            },
            "roofline": {
                "kernel": "k_linearize",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "launches": stats["linearize_launches"],
                "avg_launch_ms": lin_ms / launches,
                "algorithmic_bytes_per_launch": alg_bytes / launches,
                "algorithmic_unit": "SURVEY 8d materialised Jacobian pass: 212 B/observation + 84 B/depth observation",
                "achieved_on_stored_bytes": STORED_BYTES_PER_OBS * lin_obs / (lin_ms * 1e-3) / 1e9 if lin_ms > 0 else 0.0,
                "kernel_share_of_device_time": lin_ms / stats["total_ms"] if stats["total_ms"] > 0 else None,
            },
        }
        out["roofline_schur"] = {
            "kernel": "k_schur",
            "bound": "mfma",
            "achieved": schur_tf,
            "peak": F64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": schur_tf / F64_MFMA_PEAK_TFLOPS,
            "traffic": None,
            "launches": stats["schur_launches"],
            "avg_launch_ms": schur_ms / max(1, stats["schur_launches"]),
            "algorithmic_flops_per_launch": schur_flops / max(1, stats["schur_launches"]),
            "kernel_share_of_device_time": schur_ms / stats["total_ms"] if stats["total_ms"] > 0 else None,
        }
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(base, opts, args.cpu_windows)
        print(json.dumps(out))
    batch.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(base, opts, n_windows):
    """The oracle (Ceres-1.13 restatement, 'port') timed on this box's host cores on a bounded sample of the same
    workload: the same windows, the same schedule.  3 threads mirror the reference's opt.num_threads = 3
    (bundle_adjuster_keyframes.cpp:764)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle

    pyoracle.load()
    n = min(n_windows, len(base))
    res = {}
    for threads in (3, 1):
        t0 = time.perf_counter()
        k = 0
        for i in range(n if threads == 3 else max(1, n // 4)):
            w = base[i].copy()
            pyoracle.solve(w, opts, threads, 1)
            k += 1
        res[threads] = k / (time.perf_counter() - t0)
    return {
        "value": res[3],
        "unit": "windows/s",
        "cores": 3,
        "kind": "port",
        "sample": "%d of the benchmark's windows solved one after another by the oracle with 3 evaluation threads (Ceres num_threads=3); single-thread rate %.3f windows/s; host has %d cores" % (n, res[1], os.cpu_count()),
    }


if __name__ == "__main__":
    main()
