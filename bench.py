#!/usr/bin/env python3
"""Benchmark of the keyframe-BA hot path: window solves/sec on synthetic KITTI-00-shaped windows (BASELINE.json
configs[1]: 5 keyframes x ~2000 landmarks, LiDAR depth + ground plane).

A "step" = one pass of the hot path over one batch: reset the HBM-resident batch to its initial parameters and run the
full solveTrimmed schedule ({2 LM iterations -> quantile trimming -> solve to tolerance}) for every window of the batch
through the C-ABI (limo_ba_batch_reset + limo_ba_batch_solve).  Inputs are resident in HBM before the timed region.
value = windows solved per second over all ranks.  The windows of a batch stream through the library's slots
(kba_kernels.hip:k_sched), so the batch is several times the number of windows in flight.

  python bench.py --gpus N --steps K --warmup W [--batch B] [--no-extras] [--no-cpu-baseline] [--no-pmc]
N > 1: one rank per GPU (launched by the driver with torch.distributed.run, or by this script itself when it is
started as a plain `python bench.py --gpus N`); windows are independent, so ranks take disjoint batches with NO
data-path collective ("replicas only" for this configuration) - weak scaling.  The landmark-sharded solve
of ONE large window (configs[3], one RCCL all-gather of a 42 KB block per LM iteration) is timed in the `c4` entry when N >= 2.

Besides the headline the JSON line carries (rank 0): `roofline` (k_lin_lm, the Jacobian evaluation: `achieved` on the FUSED unit of
SURVEY 8d - never above the peak -, `traffic` measured IN this run by rocprofv3 counter passes in a child process when rocprofv3 is
on PATH (scripts/pmc_collect.py; --no-pmc skips them), `frac` = traffic / kernel time / 8 TB/s, `limited_by`), `roofline_schur` (MFMA),
`roofline_evaluate` (the MATERIALISED Jacobian kernel of SURVEY 8d), `parity` (the oracle on windows of this very batch against the GPU
results), `cpu_baseline`, `batch_sizes`, `single_window`, `end_to_end`, `depth_c3`, `c4`.
"""
import argparse
import json
import os
import socket
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
F64_MFMA_PEAK_TFLOPS = 78.6  # fp64 matrix peak: 256 CU x 4 SIMD x 32 FLOP/clk x 2.4 GHz (v_mfma_f64_16x16x4 = 2048 FLOP / 64 clk)
# Work unit of the Jacobian evaluation.  SURVEY §8d gives two: the MATERIALISED pass (212 B per observation + 84 B per depth
# observation: residual, J_pose, J_point written out) - that is k_evaluate, reported as `roofline_evaluate` - and the FUSED
# normal-equation path: 52 B read per observation + 72 B written per landmark (V, g), "+ W only if W is materialised".
# k_lin_lm is the fused path; instead of W (144 B per pair) the design stores the four scalars of the factored Jacobian
# (DESIGN.md §3): two of them, 16 B per observation, since round 5 (xn, yn are rebuilt by the readers from the landmark and
# the view).  Its algorithmic unit is therefore 52 + 16 B per observation + 72 B per landmark;
# `roofline.achieved` = that x what one launch linearises / launch time, `roofline.frac` = achieved / 8 TB/s, `roofline.traffic` = the
# HBM bytes the counters see per launch and `roofline.traffic_frac` = traffic / time / 8 TB/s (round 5 printed that one as `frac`).
FUSED_BYTES_PER_OBS = 52 + 16
FUSED_BYTES_PER_LANDMARK = 72
ACCEPT_MOVE_BYTES_PER_LANDMARK = 24  # the accepted step's landmark written into place by the relinearisation (k_accept's job until round 5)
# The kernel is bound by the fp64 pipe, not by HBM, so the line also prices it there: `roofline.fp64` =
# ALGORITHMIC fp64 flops / kernel time / 78.6 TF.  Per (landmark, view) pair the fused path is priced at ~300 multiply-adds whatever the
# code looks like (the round-6 kernel executes ~240: its camera-side sums never form the 3 x 6 pose Jacobian - the count stays the review's, so
# that the figure compares across rounds): pose + landmark Jacobian of the three residual rows ~100, the Gram terms U (21), g_c (6), V (6), g_l (3) at three
# rows each ~110 + the 28-value cross-lane sum they leave through ~50, robust loss + corrector ~20, one reciprocal and two inverse
# square roots ~20 - the count the round-4 review derived from cost_functors_ceres.hpp:71-155,193-212.
FP64_FLOPS_PER_LINEARISED_OBS = 2 * 300
def _newest_profile(suffix):
    import glob

    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return found[-1] if found else os.path.join(ROOT, "profiles", "r02_" + suffix)


PMC_FILE = _newest_profile("pmc_kernels.json")
PMC_DEPTH_FILE = _newest_profile("pmc_depth_kernels.json")


def kernel_source_sha16():
    """sha256 (16 hex digits) of the BA kernel sources: a stored PMC profile is only used for the file it was taken on."""
    import hashlib

    h = hashlib.sha256()
    for name in ("kba_kernels.hip", "kba_items.hpp", "kba_math.hpp", "kba_layout.hpp", "kba_lm.hpp"):
        with open(os.path.join(ROOT, "limo_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]
C4_GRACE_S = 300
UNFUSED_PMC_FILE = os.path.join(ROOT, "profiles", "r02_pmc_kernels_unfused_pipeline.json")
UNFUSED_BYTES_PER_OBS = 226.1  # k_linearize 101.3 + k_lm_accum 92.6 + k_lm_damp 32.2 B per observation (that file), before they became one pass


def _make_window(args):
    from limo_amd import synth

    seed, n_kf, n_lm = args
    return synth.make_window(seed, n_kf=n_kf, n_lm=n_lm)


def generate_windows(seeds, n_kf, n_lm):
    """Synthetic windows, generated by a pool of host processes (10 ms each in numpy; forked before any GPU state exists)."""
    jobs = [(s, n_kf, n_lm) for s in seeds]
    n_proc = min(32, os.cpu_count() or 1, max(1, len(jobs) // 16))
    if n_proc <= 1:
        return [_make_window(j) for j in jobs]
    import multiprocessing as mp

    with mp.get_context("fork").Pool(n_proc) as pool:
        return pool.map(_make_window, jobs, chunksize=16)


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
        return [{"num_linearizations": 1, "iterations_total": 1, "successful_steps": 1, "n_trimmed_landmarks": 0, "termination": 0, "final_cost": 1.0} for _ in self.windows]

    def close(self):
        pass


def relaunch_with_ranks(n):
    """`python bench.py --gpus N` started as ONE process: become N ranks (one per GPU) under torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("LIMO_BENCH_BATCH", "16384")), help="independent windows per GPU per step")
    ap.add_argument("--n-kf", type=int, default=5)
    ap.add_argument("--n-lm", type=int, default=2000)
    ap.add_argument("--distinct", type=int, default=int(os.environ.get("LIMO_BENCH_DISTINCT", "4096")),
                    help="distinct windows generated per rank (0 = every window of the batch is different); the batch repeats them, "
                         "every copy is an independent solve on the device")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline + rooflines only (no parity / single-window / end-to-end / depth / C4 entries)")
    ap.add_argument("--cpu-solves", type=int, default=20)
    ap.add_argument("--no-pmc", action="store_true", help="do not run the rocprofv3 counter passes (roofline.traffic then comes from the stored profile, if its kernel sha matches)")
    ap.add_argument("--pmc-timeout", type=float, default=240.0, help="seconds the counter passes may take altogether")
    ap.add_argument("--selftest-dist", action="store_true",
                    help="CPU check of the multi-rank plumbing only (rendezvous over gloo, barriers, max-over-ranks timing, "
                         "rank-0 JSON): the solve is replaced by a stub, nothing is measured (tests/test_bench_dist.py)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        relaunch_with_ranks(args.gpus)  # does not return
    selftest = args.selftest_dist
    if selftest:
        args.batch, args.n_lm, args.no_cpu_baseline, args.no_extras = min(args.batch, 4), min(args.n_lm, 60), True, True

    # ---- synthetic input first (host processes; no GPU state yet): `distinct` different windows per rank, tiled to the
    # batch size (each copy is an independent solve), distinct seeds per rank
    distinct = args.batch if args.distinct <= 0 else max(1, min(args.distinct, args.batch))
    base = generate_windows([1000 + 100000 * rank + i for i in range(distinct)], args.n_kf, args.n_lm)
    extra_windows = None
    if rank == 0 and world == 1 and not args.no_extras and not selftest:  # inputs of the end-to-end / evaluate entries
        extra_windows = generate_windows([7000 + i for i in range(1024)], args.n_kf, args.n_lm)

    import numpy as np
    import torch

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

    from limo_amd import default_options

    opts = default_options()
    ctx = None
    if not selftest:
        from limo_amd import ba

        ctx = ba.Context(local_rank)
    # (copies of a window share its host buffers: identical solves write identical results back)
    windows = [base[i % distinct] for i in range(args.batch)]
    pristine = [w.copy() for w in base[: max(32, args.cpu_solves)]]  # untouched inputs for the CPU baseline / parity / extras
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
    reps = batch.download()  # batch.windows now hold the GPU results
    # Per-kernel times for the rooflines: one more step with ONE slot group - with several groups the kernels of
    # different groups overlap on the device and event timing of a single kernel means nothing.
    if not selftest:
        os.environ["KBA_GROUPS"] = "1"
        batch.kernel_stats(reset=True)
        step()
        torch.cuda.synchronize()
        del os.environ["KBA_GROUPS"]
    stats = batch.kernel_stats(reset=False)
    prof_steps = 1 if not selftest else args.steps

    out = None
    if rank == 0:
        total_windows = args.batch * world * args.steps
        value = total_windows / elapsed
        # ---- roofline of the Jacobian evaluation (k_lin_lm): what one launch linearises, from the reports of this very run
        lin_obs = sum(r["num_linearizations"] * w.n_obs for r, w in zip(reps, windows)) * prof_steps
        lin_lms = sum(r["num_linearizations"] * w.n_lm for r, w in zip(reps, windows)) * prof_steps
        # (round 6: the kernel also moves the landmarks of an accepted step from the candidate buffer into place - the former k_accept
        # pass -: 24 B written per landmark and accepted step; every accepted step is followed by exactly one linearisation)
        acc_lms = sum(r["successful_steps"] * w.n_lm for r, w in zip(reps, windows)) * prof_steps
        alg_bytes = FUSED_BYTES_PER_OBS * lin_obs + FUSED_BYTES_PER_LANDMARK * lin_lms + ACCEPT_MOVE_BYTES_PER_LANDMARK * acc_lms
        launches = max(1, stats["linearize_launches"])
        lin_ms = stats["linearize_ms"]
        lin_s = max(1e-12, lin_ms * 1e-3)
        achieved = alg_bytes / lin_s / 1e9
        # ---- second dominant kernel: the Schur complement (f64 MFMA).  Algorithmic flops of one window iteration =
        # n_c^2 * 3N (SYRK of the 3N x n_c landmark-eliminated block; n_c free camera slots, N landmarks left)
        nf = [10 * (w.n_kf - 1) for w in windows]  # first keyframe Pose-fixed, every other keyframe 6 + 3 + 1 slots
        schur_flops = sum(r["iterations_total"] * (n * n * 3.0 * (w.n_lm - r["n_trimmed_landmarks"])) for r, n, w in zip(reps, nf, windows)) * prof_steps
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
                # share of the LM iterations whose step was accepted (what a speculative linearisation at the candidate would
                # have to beat: DESIGN.md §4)
                "accept_ratio": float(sum(r["successful_steps"] for r in reps)) / max(1, sum(r["iterations_total"] for r in reps)),
                "linearisations_per_iteration": float(sum(r["num_linearizations"] for r in reps)) / max(1, sum(r["iterations_total"] for r in reps)),
            },
            "roofline": {
                "kernel": "k_lin_lm (Jacobian evaluation + landmark blocks + damping, landmark-major; SURVEY 8d's fused normal-equation path)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "frac_definition": "achieved / peak: algorithmic bytes / kernel time / 8 TB/s (the fraction on the bytes the counters saw: traffic_frac)",
                "traffic": None,
                "traffic_source": None,
                "launches": stats["linearize_launches"],
                "avg_launch_ms": lin_ms / launches,
                "algorithmic_bytes_per_launch": alg_bytes / launches,
                "algorithmic_unit": "SURVEY 8d fused path: 52 B read per observation + 72 B (V, g) written per landmark, + the 16 B per observation of factored "
                                    "Jacobian planes this design stores in place of W (144 B per pair), + 24 B per landmark and ACCEPTED step (the landmark moved into place by the "
                                    "relinearisation since round 6); x the observations / landmarks this run linearised",
                "fp64": {"achieved": FP64_FLOPS_PER_LINEARISED_OBS * lin_obs / lin_s / 1e12, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": FP64_FLOPS_PER_LINEARISED_OBS * lin_obs / lin_s / 1e12 / F64_MFMA_PEAK_TFLOPS,
                         "algorithmic_flops_per_observation": FP64_FLOPS_PER_LINEARISED_OBS,
                         "what": "algorithmic fp64 flops of the fused linearisation (~300 multiply-adds per (landmark, view) pair) / kernel time / the fp64 vector = matrix peak"},
                "linearised_observations_per_launch": lin_obs / launches,
                "measured_in": "one extra step with a single slot group (kernels of different groups overlap in the timed steps); HIP events on the kernel's stream",
                "materialised_kernel": "the SURVEY 8d MATERIALISED Jacobian pass (212 B / observation + 84 B / depth observation) is k_evaluate: see roofline_evaluate",
                "kernel_share_of_device_time": lin_ms / stats["total_ms"] if stats["total_ms"] > 0 else None,
                "_lin_obs": lin_obs, "_lin_s": lin_s,
            },
            "roofline_schur": {
                "kernel": "k_schur_lean (plain + ground-plane blocks)",
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
            },
        }
    batch.close()

    # ---- HBM traffic of k_lin_lm, MEASURED in this run when rocprofv3 is there: two counter passes (FETCH_SIZE, WRITE_SIZE - they
    # do not share a pass) + one SQ pass over full 1024-window rounds of the same kernels (scripts/pmc_collect.py), in a child
    # process after the batch has released its memory; else the stored, sha-bound profile; else nothing.
    if rank == 0 and not selftest:
        fill_traffic(out["roofline"], out["roofline_schur"], measure=not args.no_pmc and world == 1, timeout=args.pmc_timeout)
    elif rank == 0:
        out["roofline"].pop("_lin_obs"), out["roofline"].pop("_lin_s")

    # ---- C4: landmark-sharded solve of one large window (every rank takes part when N >= 2)
    c4 = None
    if not args.no_extras and not selftest:
        # The sharded C4 solve is the one place where the ranks exchange data over RCCL inside the library.  The headline
        # above is complete at this point: should that exchange ever stall on a node this code has not seen (it is covered
        # by a one-rank communicator on the GPU and by a two-rank gloo test of the same exchange logic), every rank leaves
        # after a grace period and rank 0 still prints the line, with the failure recorded instead of a number.
        watchdog = None
        if world >= 2:
            import threading

            def bail():
                if rank == 0:
                    out["c4"] = {"error": "landmark-sharded C4 solve did not finish within %d s" % C4_GRACE_S}
                    print(json.dumps(out), flush=True)
                os._exit(0)

            watchdog = threading.Timer(C4_GRACE_S, bail)
            watchdog.daemon = True
            watchdog.start()
        try:
            c4 = bench_c4(ctx, dist, rank, world, opts)
        except Exception as e:  # the headline must not depend on this extra
            c4 = {"error": "%s: %s" % (type(e).__name__, e)}
        if watchdog is not None:
            watchdog.cancel()
    def extra(fn, *a):  # the headline and the rooflines above are complete: an extra that fails is recorded, not fatal
        try:
            return fn(*a)
        except Exception as e:
            return {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0 and not selftest and not args.no_extras:
        out["c4"] = c4
        if world == 1:
            out["roofline_evaluate"] = extra(bench_evaluate, ctx, extra_windows, opts)
            out["single_window"] = extra(bench_single_window, ctx, pristine, opts)
            out["batch_sizes"] = extra(bench_batch_sizes, ctx, extra_windows, opts)
            out["end_to_end"] = extra(bench_end_to_end, ctx, extra_windows, opts)
            out["depth_c3"] = extra(bench_depth, ctx)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not selftest:  # reported on rank 0 at N = 1 only
        res = extra(cpu_baseline_and_parity, pristine, windows, reps, opts, args.cpu_solves)
        out["cpu_baseline"], out["parity"] = res if isinstance(res, tuple) else (res, res)
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------- extras
def fill_traffic(roof, roof_schur, measure, timeout):
    """roofline.traffic / frac / limited_by from counters: measured now (scripts/pmc_collect.py under rocprofv3), or from the
    newest stored profile if it was taken on these very kernel sources."""
    import shutil
    import subprocess
    import tempfile

    lin_obs, lin_s = roof.pop("_lin_obs"), roof.pop("_lin_s")
    pmc, src = None, None
    if measure and shutil.which("rocprofv3"):
        with tempfile.NamedTemporaryFile(suffix=".json", dir="/tmp") as tf:
            t0 = time.perf_counter()
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_collect.py"), "--out", tf.name, "--passes", "hbm,valu", "--timeout", str(timeout),
                                    "--per-pass", str(max(5.0, timeout / 3.0))], capture_output=True, text=True, timeout=timeout + 30)
                if r.returncode == 0:
                    pmc = json.load(open(tf.name))
                    src = "measured in this run (%.0f s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over full 1024-window rounds (scripts/pmc_collect.py); FETCH_SIZE x2 on gfx950" % (time.perf_counter() - t0)
                    if pmc.get("failed_passes"):  # (a pass that failed or hung is skipped by pmc_collect; what the others measured counts)
                        roof["traffic_measurement_partial"] = "; ".join(pmc["failed_passes"])[-300:]
                else:
                    roof["traffic_measurement_failed"] = (r.stderr or r.stdout)[-200:]
            except Exception as e:  # noqa: BLE001 - the line must not depend on the profiler
                roof["traffic_measurement_failed"] = "%s: %s" % (type(e).__name__, e)
    def lin_of(p):
        return next((v for k, v in sorted(p.get("kernels", {}).items()) if k.startswith("k_lin_lm")), {}) if p else {}

    stored = json.load(open(PMC_FILE)) if os.path.exists(PMC_FILE) else None
    stored_ok = stored is not None and stored.get("kernel_source_sha16") == kernel_source_sha16()
    if pmc is not None and "hbm_bytes_per_observation" not in lin_of(pmc):
        pmc = None  # (the HBM passes of this run failed: the stored profile, if it is of these sources)
    if pmc is not None and stored_ok and lin_of(pmc).get("valu_busy") is None and lin_of(stored).get("valu_busy") is not None:
        # the SQ pass of this run failed: its ratios from the stored profile of the same kernel sources
        for k, v in stored.get("kernels", {}).items():
            for key in ("valu_busy", "lds_bank_conflict_per_lds_inst"):
                if key in v and k in pmc["kernels"] and key not in pmc["kernels"][k]:
                    pmc["kernels"][k][key] = v[key]
        src += "; valu_busy / LDS conflict ratio from the stored profile %s (same kernel sources)" % os.path.relpath(PMC_FILE, ROOT)
    if pmc is None and stored is not None:
        if stored_ok:
            pmc = stored
            src = "stored profile %s (taken on these kernel sources: sha %s); FETCH_SIZE x2 on gfx950" % (os.path.relpath(PMC_FILE, ROOT), stored.get("kernel_source_sha16"))
        else:
            roof["traffic_source"] = "none: no rocprofv3 in this run and %s was taken on other kernel sources (sha %s, now %s)" % (
                os.path.relpath(PMC_FILE, ROOT), stored.get("kernel_source_sha16"), kernel_source_sha16())
    if pmc is None:
        return
    kern = pmc.get("kernels", {})
    lin = next((v for k, v in sorted(kern.items()) if k.startswith("k_lin_lm") and "hbm_bytes_per_observation" in v), None)
    if lin:
        per_obs = lin["hbm_bytes_per_observation"]
        roof["traffic"] = per_obs * lin_obs / max(1, roof["launches"])
        roof["traffic_bytes_per_observation"] = per_obs
        # (`frac` stays achieved / peak on the ALGORITHMIC bytes - the contract of the line; the fraction on what moved has its own name)
        roof["traffic_frac"] = per_obs * lin_obs / lin_s / 1e9 / HBM_PEAK_GBPS
        roof["traffic_frac_definition"] = "HBM bytes that moved (counters: per linearised observation of a full round x this run's linearised observations) / this run's kernel time / 8 TB/s"
        roof["traffic_source"] = src
        roof["traffic_over_algorithmic"] = roof["traffic"] / roof["algorithmic_bytes_per_launch"]
        if lin.get("valu_busy") is not None:
            roof["limited_by"] = {"unit": "fp64-valu", "valu_busy": lin["valu_busy"],
                                  "definition": "4 x SQ_ACTIVE_INST_VALU / 32 SIMDs of the sampled shader engine / SQ_BUSY_CYCLES: the share of cycles the vector port issues - "
                                                "above ~0.75 the kernel is bound by fp64 issue, not by HBM"}
    if pmc.get("round_hbm_bytes_per_observation"):
        roof["whole_iteration_hbm_bytes_per_observation"] = pmc["round_hbm_bytes_per_observation"]
    sch = [v for k, v in kern.items() if k.startswith("k_schur_lean")]
    if sch and all("lds_bank_conflict_per_lds_inst" in v for v in sch):
        roof_schur["lds_bank_conflict_per_lds_inst"] = max(v["lds_bank_conflict_per_lds_inst"] for v in sch)


def _median_ms(fn, n, warm):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(1e3 * (time.perf_counter() - t0))
    return statistics.median(ts), ts


def bench_single_window(ctx, base, opts):
    """What LIMO itself does: ONE window per call, host buffers in and out (limo_ba_solve = pack + upload + solve +
    download), windows taken in turn."""
    ws = base[:20]
    it = iter(range(10**9))

    def one():
        ctx.solve(ws[next(it) % len(ws)].copy(), opts)

    med, ts = _median_ms(one, 20, 3)
    # the other per-frame call of the reference's node: adjustPoseOnly of the new frame against the fixed landmarks
    from limo_amd import default_options, synth

    pw, prior, _ = synth.make_pose_only_case(71)
    po = default_options(min_landmarks_for_trimming=30)
    pmed, pts = _median_ms(lambda: ctx.adjust_pose_only(pw.copy(), prior, po), 50, 5)
    return {"value": 1e3 / med, "unit": "windows/s", "ms_per_solve_median": med, "ms_min": min(ts), "ms_max": max(ts),
            "what": "limo_ba_solve on C2 windows, one per call, host buffers (pack + upload + solve + download per call); one cooperative launch per call (k_solve_coop)",
            "adjust_pose_only": {"ms_per_call_median": pmed, "ms_min": min(pts), "landmarks": int(pw.n_lm), "observations": int(pw.n_obs),
                                 "what": "limo_ba_adjust_pose_only, one launch per call (k_solve_wg)"}}


def bench_batch_sizes(ctx, base, opts):
    """SURVEY 8d's batch set B in {1, 64, 1024}: headline-style lines (reset + full solveTrimmed schedule of an HBM-resident
    batch) at the small batch sizes; B = 1 is what the reference pipeline itself issues (<= 2.55 solves/s)."""
    from limo_amd import ba

    out = {}
    for B, steps in ((1, 20), (64, 10), (1024, 4)):
        b = ba.Batch(ctx, [w.copy() for w in base[:B]])
        b.reset()
        b.solve(opts)
        ts = []
        for _ in range(steps):
            t0 = time.perf_counter()
            b.reset()
            b.solve(opts)
            ts.append(time.perf_counter() - t0)
        b.close()
        med = statistics.median(ts)
        out[str(B)] = {"value": B / med, "unit": "windows/s", "ms_per_step": 1e3 * med, "steps": steps,
                       "what": "reset + solve of %d HBM-resident C2 windows (inputs resident, no upload / download in the timed region)" % B}
    return out


def bench_end_to_end(ctx, base, opts):
    """Host flattening included: limo_ba_batch_create (pack + upload) + solve + download of 1024 windows handed over as
    host buffers (SURVEY §8d "end-to-end" next to the HBM-resident "kernel" number).  The array of limo_ba_window structs
    is made before the clock starts (a C / C++ caller holds it anyway; ctypes needs ~20 us of Python per window for it).
    (Tried and dropped: a stream of such batches through TWO contexts, batch k + 1 packed and uploaded by a second host thread
    while batch k solves - 9.9 k windows/s against 11.3 k for the serial calls on the same box: the runtime serialises the
    pageable upload of one thread with the launches of the other.)"""
    from limo_amd import ba
    from limo_amd.window import struct_array

    ws = [w.copy() for w in base[:1024]]
    b = ba.Batch(ctx, [w.copy() for w in ws])
    b.solve(opts)
    b.download()  # (warm-up of the whole path: the context's pinned result buffer is allocated once, here)
    b.close()
    arr = struct_array(ws)
    t0 = time.perf_counter()
    b = ba.Batch(ctx, ws, arr)
    t1 = time.perf_counter()
    b.solve(opts)
    t2 = time.perf_counter()
    b.download()
    t3 = time.perf_counter()
    b.close()
    out = {"value": len(ws) / (t3 - t0), "unit": "windows/s", "batch": len(ws), "create_ms": 1e3 * (t1 - t0), "solve_ms": 1e3 * (t2 - t1),
           "download_ms": 1e3 * (t3 - t2), "resident_windows_per_s": len(ws) / (t2 - t1),
           "what": "pack + upload + solve + download of host windows (PCIe-inclusive); never the headline value"}
    return out


def bench_evaluate(ctx, base, opts):
    """The MATERIALISED Jacobian evaluation SURVEY §8d grades (k_evaluate behind limo_ba_evaluate: residuals 3 + J_pose
    18 + J_point 9 doubles per observation written), timed on the device over a full batch."""
    from limo_amd import ba

    ws = base[:1024]
    ms, n_obs, n_dep = ba.evaluate_batch_time(ctx, ws, opts, reps=10)
    alg = 212 * n_obs + 84 * n_dep  # SURVEY 8d, materialised pass
    # what the round-6 kernel moves: read u, v, d, landmark index 16 B + landmark 24 B + weight 8 B; write the rows that exist
    # (rows u, v: 20 doubles per observation; depth row: 10 doubles per DEPTH observation) + 1 validity byte
    moved = (48 + 8 * 20 + 1) * n_obs + 8 * 10 * n_dep
    return {"kernel": "k_view_consts_all + k_evaluate", "bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None, "launch_ms": ms, "observations": n_obs,
            "algorithmic_unit": "SURVEY 8d: 212 B/observation + 84 B/depth observation",
            "achieved_on_bytes_written_and_read": moved / (ms * 1e-3) / 1e9, "batch": len(ws)}


def bench_depth(ctx):
    """BASELINE.json configs[2]: 64-beam sweeps (~112 k returns) + 1500 features per frame.  Host-inclusive: one frame per
    call (limo_depth_estimate, host cloud in / host depths out) and 32 frames per call (limo_depth_estimate_batch);
    device-resident: 32 sweeps already in HBM, depths left in HBM (the rate the roofline fraction is quoted on)."""
    import numpy as np
    import torch

    from limo_amd import ba, synth_lidar

    n_batch = 32
    frames = [synth_lidar.make_frame(1 + k) for k in range(n_batch)]
    fr = frames[0]
    d = ba.depth_estimate(ctx, fr)
    single, _ = _median_ms(lambda: ba.depth_estimate(ctx, fr), 50, 3)
    batch_host, _ = _median_ms(lambda: ba.depth_estimate_batch(ctx, frames), 10, 2)
    dev = []
    for f in frames:
        g = dict(f)
        g["cloud"] = torch.from_numpy(np.ascontiguousarray(f["cloud"], np.float32)).cuda()
        g["uv"] = torch.from_numpy(np.ascontiguousarray(f["uv"], np.float32)).cuda()
        g["is_ground"] = torch.from_numpy(np.ascontiguousarray(f["is_ground"], np.uint8)).cuda()
        dev.append(g)
    batch_dev, _ = _median_ms(lambda: ba.depth_estimate_batch(ctx, dev, device=True), 20, 3)
    # device time per kernel of the same 32-frame call, HIP events on the library's stream
    ctx.lib.limo_depth_set_timing(ctx.ptr, 1)
    kms = []
    for _ in range(10):
        ba.depth_estimate_batch(ctx, dev, device=True)
        kms.append(ba.depth_kernel_ms(ctx))
    ctx.lib.limo_depth_set_timing(ctx.ptr, 0)
    k_ms = {k: statistics.median(x[k] for x in kms) for k in kms[0]}
    n_pts = sum(f["cloud"].shape[0] for f in frames)
    n_feat = sum(f["uv"].shape[0] for f in frames)
    vis = sum(synth_lidar.visible_points(f) for f in frames)
    alg_d1 = 16 * n_pts + 20 * vis  # 16 B / return read + 20 B / visible return written
    alg_feat = 212 * n_feat         # 212 B / feature
    alg = alg_d1 + alg_feat
    # counter traffic of the depth kernels: only from a stored profile that was taken on THIS depth.hip (the file carries the sha of
    # the source it ran on; scripts/gpu_depth_prof.sh PMC=1 writes it)
    pmc_d, pmc_note = {}, "no stored counter profile of the depth kernels"
    if os.path.exists(PMC_DEPTH_FILE):
        import hashlib

        with open(PMC_DEPTH_FILE) as f:
            stored = json.load(f)
        with open(os.path.join(ROOT, "limo_amd", "csrc", "depth.hip"), "rb") as f:
            sha = hashlib.sha256(f.read()).hexdigest()[:16]
        if stored.get("depth_source_sha16") == sha:
            pmc_d = stored.get("kernels", {})
        else:
            pmc_note = "none: %s was taken on another depth.hip (sha %s, now %s)" % (os.path.relpath(PMC_DEPTH_FILE, ROOT), stored.get("depth_source_sha16"), sha)

    def roof(kernel, alg_bytes, ms):
        tr = next((v.get("hbm_MB") for k, v in pmc_d.items() if k.startswith(kernel) and v.get("frames") == n_batch), None)
        return {"kernel": kernel, "bound": "hbm", "achieved": alg_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "launch_ms": ms, "algorithmic_bytes_per_launch": alg_bytes,
                "traffic": None if tr is None else tr * 1e6,
                "traffic_source": pmc_note if tr is None else "%s (stored rocprofv3 --pmc passes of a %d-frame call, taken on this depth.hip)" % (os.path.relpath(PMC_DEPTH_FILE, ROOT), n_batch)}

    return {"value": 1e3 / single, "unit": "frames/s", "ms_per_frame_median": single,
            "points": fr["cloud"].shape[0], "features": fr["uv"].shape[0], "visible_points": synth_lidar.visible_points(fr),
            "features_with_depth": int((d > 0).sum()),
            "what": "limo_depth_estimate, one frame per call, host cloud in / host depths out (host-inclusive), 7 kernel launches",
            "batch_of_32_host": {"frames_per_s": 1e3 * n_batch / batch_host, "ms_per_frame": batch_host / n_batch,
                                 "what": "limo_depth_estimate_batch, 32 host sweeps per call (PCIe-inclusive), 7 kernel launches per call"},
            "batch_of_32_device_resident": {"frames_per_s": 1e3 * n_batch / batch_dev, "ms_per_frame": batch_dev / n_batch,
                                            "algorithmic_bytes_per_frame": alg / n_batch,
                                            "hbm_frac_on_algorithmic_bytes": alg / (batch_dev * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                            "what": "sweeps and features resident in HBM, depths left in HBM; host time of the call incl. its 7 launches and the final sync"},
            "roofline_depth": {"measured_with": "HIP events on the library's stream around the kernels of one 32-frame call (limo_depth_set_timing), median of 10 calls",
                               "kernel_ms": k_ms,
                               "k_project": roof("k_project", alg_d1, k_ms["k_project"]),
                               "k_features": roof("k_features", alg_feat, k_ms["k_features"]),
                               "all_kernels": {"achieved": alg / (k_ms["total"] * 1e-3) / 1e9, "unit": "GB/s", "frac": alg / (k_ms["total"] * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                               "algorithmic_unit": "SURVEY 8d: D1 16 B/return + 20 B/visible return, D2-D5 212 B/feature"}}}


def bench_c4(ctx, dist, rank, world, opts):
    """BASELINE.json configs[3]: 10 keyframes x 8000 landmarks.  One GPU: plain solve; N >= 2 ranks: landmarks sharded
    over the ranks, RCCL all-reduce of the partial camera blocks per LM iteration (limo_ba_solve_sharded)."""
    import numpy as np

    from limo_amd import dist as ldist, synth

    c4 = synth.config_c4()
    out, res = None, {}
    if rank == 0:

        def one():
            w = c4.copy()
            res["rep"] = ctx.solve(w, opts)
            res["w"] = w

        med, _ = _median_ms(one, 5, 1)
        out = {"ms_one_gpu": med, "lm_iterations": res["rep"]["iterations_total"], "observations": c4.n_obs, "final_cost": res["rep"]["final_cost"]}
        if world == 1:  # the exchange accounting of the sharded solve, on 4 virtual shards (no communicator: the unpack only)
            rs = ctx.solve_sharded(c4.copy(), opts, 4)
            st = ctx.exchange_stats()
            # per SHARD (4 virtual shards on this GPU: every exchange point is counted once per shard)
            out["virtual_4_shards"] = {"exchange_steps_per_shard": st["exchanges"] / 4, "lm_iterations": st["iterations"],
                                       "exchange_steps_per_iteration_and_shard": st["exchanges"] / 4 / max(1, st["iterations"]),
                                       "bytes_sent_per_iteration_and_shard": st["bytes"] / 4 / max(1, st["iterations"]),
                                       "what": "per LM iteration a shard contributes ONE block (camera-side sums, ground-plane blocks, [S | rhs] upper triangle: ~41 KB) "
                                               "before camera assembly + solve and nine doubles before the step decision; all-gather, nothing summed on the wire",
                                       "rel_cost_diff_vs_unsharded": abs(rs["final_cost"] - out["final_cost"]) / abs(out["final_cost"])}
    if world >= 2:
        n_shards = world
        ldist.init_shard_comm(dist, ctx)
        ctx.solve_sharded(c4.copy(), opts, n_shards)
        dist.barrier()
        ts = []
        for _ in range(5):
            w = c4.copy()
            t0 = time.perf_counter()
            rs = ctx.solve_sharded(w, opts, n_shards)
            ts.append(1e3 * (time.perf_counter() - t0))
        ms = ldist.max_over_ranks(dist, statistics.median(ts), device="cuda")
        st = ctx.exchange_stats()
        if rank == 0:
            out.update({"ms_sharded": ms, "n_shards": n_shards, "exchange": "RCCL ncclAllGather of the shards' blocks over %d ranks (+ one ncclAllReduce per trimming round and for the landmarks)" % world,
                        "collective_calls": st["exchanges"], "collective_calls_per_iteration": st["exchanges"] / max(1, st["iterations"]),
                        "bytes_sent_per_call": st["bytes"] / max(1, st["exchanges"]),
                        "rel_cost_diff_vs_one_gpu": abs(rs["final_cost"] - out["final_cost"]) / abs(out["final_cost"]),
                        "max_pose_diff_vs_one_gpu": float(np.abs(w.kf_pose - res["w"].kf_pose).max())})
    return out


def cpu_baseline_and_parity(base, windows, reps, opts, n_solves):
    """The oracle (Ceres-1.13 restatement, kind 'port' - Ceres itself cannot be built in this image) timed on this box's
    host cores on a bounded sample of the benchmark's own windows, same schedule: 3 warm-ups, median of n_solves solves
    with 3 evaluation threads (the reference's opt.num_threads = 3, bundle_adjuster_keyframes.cpp:764), a second pass
    with all cores (capped at 64 OpenMP threads), and the phase split a Ceres FullReport would show.  The same oracle
    results are compared with what the GPU produced for those windows in the timed batch (`parity`)."""
    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle

    pyoracle.load()
    cores = os.cpu_count() or 1
    n = min(n_solves, len(base))
    for i in range(3):
        pyoracle.solve(base[i % len(base)].copy(), opts, 3, 1)
    ts, phases, worst_c, worst_p = [], np.zeros(4), 0.0, 0.0
    for i in range(n):
        w = base[i].copy()
        t0 = time.perf_counter()
        ro, pt = pyoracle.solve(w, opts, 3, 1)
        ts.append(time.perf_counter() - t0)
        phases += pt
        g, rg = windows[i], reps[i]  # window i of the batch is a copy of base[i]
        worst_c = max(worst_c, abs(rg["final_cost"] - ro["final_cost"]) / abs(ro["final_cost"]))
        worst_p = max(worst_p, float(np.abs(g.kf_pose[:, 4:] - w.kf_pose[:, 4:]).max() / max(1e-12, np.abs(w.kf_pose[:, 4:]).max())))
    med3 = statistics.median(ts)
    nt = min(cores, 64)
    ta = []
    for i in range(min(8, n)):
        w = base[i].copy()
        t0 = time.perf_counter()
        pyoracle.solve(w, opts, nt, nt)
        ta.append(time.perf_counter() - t0)
    # whole-host throughput: floor(cores / 3) solves at a time, 3 evaluation threads each (a batched-throughput headline is
    # compared with what the host can do with ALL its cores on independent windows, not with one solve spread thin)
    from concurrent.futures import ThreadPoolExecutor

    n_par = max(1, cores // 3)
    per_worker = 2
    jobs = [base[i % len(base)].copy() for i in range(n_par * per_worker)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=n_par) as ex:  # ctypes releases the GIL: the solves run concurrently
        list(ex.map(lambda w: pyoracle.solve(w, opts, 3, 1), jobs))
    host_s = time.perf_counter() - t0
    cpu = {
        "host_throughput": {"value": len(jobs) / host_s, "unit": "windows/s", "concurrent_solves": n_par, "threads_per_solve": 3, "host_cores": cores,
                            "solves": len(jobs), "wall_s": host_s,
                            "what": "floor(cores / 3) oracle solves in flight, 3 threads each, every hardware thread of the box busy"},
        "value": 1.0 / med3,
        "unit": "windows/s",
        "cores": 3,
        "kind": "port",
        "sample": "%d of the benchmark's windows, one after another, 3 warm-ups, median solve time %.1f ms with 3 evaluation threads (Ceres num_threads = 3)" % (n, 1e3 * med3),
        "all_cores": {"value": 1.0 / statistics.median(ta), "threads": nt, "host_cores": cores, "solves": len(ta)},
        "phase_split_3_threads": {"jacobian_evaluation": phases[0] / phases[3], "schur_eliminate_backsubstitute": phases[1] / phases[3],
                                  "cholesky": phases[2] / phases[3], "other": 1.0 - (phases[0] + phases[1] + phases[2]) / phases[3]},
        "label": "CPU baseline = Ceres-1.13 restatement (oracle/), not Ceres",
    }
    parity = {"n": n, "max_rel_cost": worst_c, "max_rel_pose": worst_p, "tolerance": 1e-4,
              "what": "oracle vs the GPU results of the same windows taken from the timed batch (final cost, keyframe translations)"}
    return cpu, parity


if __name__ == "__main__":
    main()
