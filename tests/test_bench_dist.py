"""bench.py's multi-rank plumbing on CPU: the driver launches `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N ...`; with --selftest-dist the same code path runs over gloo with a stub instead of the GPU batch, so the
rendezvous, the barriers around the timed region, the max-over-ranks time and the single rank-0 JSON line are covered
without a GPU (the real run needs N GPUs, which only the driver has)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_print_one_line_with_the_slowest_ranks_time():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--selftest-dist"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["higher_is_better"] is True
    assert "cpu_baseline" not in out  # N = 1 only
    # stub: rank r sleeps 20 (1 + r) ms per step -> the slowest rank needs >= 40 ms per step
    assert out["ms_per_step"] >= 40.0
    assert abs(out["value"] - 2 * out["config"]["batch_windows_per_gpu"] * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]
    assert "NOT a measurement" in out["data"]


def test_gpus_flag_without_launcher_starts_its_own_ranks():
    """`python bench.py --gpus N` as ONE process re-executes itself under torch.distributed.run (a driver
    that runs the documented command must get its N-GPU line)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--selftest-dist"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0])["n_gpus"] == 2


def test_counter_collection_degrades_without_a_profiler(monkeypatch):
    """scripts/pmc_collect.py (the in-run traffic measurement of bench.py): without rocprofv3 on PATH it reports why and returns
    nothing - bench.py then falls back to the stored, sha-bound profile or leaves `traffic` null; the sha it stamps is the one
    bench.py checks."""
    import importlib.util
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pmc_collect", os.path.join(root, "scripts", "pmc_collect.py"))
    pmc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pmc)
    monkeypatch.setenv("PATH", "/nonexistent")
    out, err = pmc.collect(["hbm"], 10.0)
    assert out is None and "rocprofv3" in err
    sys.path.insert(0, root)
    import bench

    assert pmc.kernel_source_sha16() == bench.kernel_source_sha16()
