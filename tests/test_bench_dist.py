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


_FAKE_ROCPROF = r'''#!/usr/bin/env python3
# a stand-in for rocprofv3 (test_counter_collection_skips_a_pass_that_hangs): counter passes write the sqlite layout
# scripts/pmc_collect.py reads; a pass that asks for SQ_BUSY_CYCLES hangs
import json, os, sqlite3, sys, time
a = sys.argv[1:]
ctrs = a[a.index("--pmc") + 1:a.index("--kernel-trace")]
d = a[a.index("-d") + 1]
if "SQ_BUSY_CYCLES" in ctrs:
    time.sleep(600)
os.makedirs(os.path.join(d, "host"), exist_ok=True)
db = sqlite3.connect(os.path.join(d, "host", "p_results.db"))
db.execute("create table pmc_events (name text, counter_name text, counter_value real, duration real)")
for c in ctrs:
    db.execute("insert into pmc_events values (?, ?, ?, ?)", ("void kba::k_lin_lm<3>(kba::BatchView)", c, 1000.0 if c == "FETCH_SIZE" else 500.0, 300e3))
db.commit()
print(json.dumps({"windows": 1024, "observations": 10000, "landmarks": 2000, "depth_observations": 4000, "free_slots": 40}))
'''


def test_counter_collection_skips_a_pass_that_hangs(tmp_path, monkeypatch):
    """A counter pass that hangs (seen once on a loaded box) is killed with its process group after its own time limit and skipped:
    the HBM passes still deliver `traffic`; bench.py takes the missing SQ ratios from the stored profile when that is of the same
    kernel sources, and says so."""
    import importlib.util
    import os
    import stat
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "bin"
    fake.mkdir()
    exe = fake / "rocprofv3"
    exe.write_text(_FAKE_ROCPROF)
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(fake) + os.pathsep + os.environ["PATH"])
    spec = importlib.util.spec_from_file_location("pmc_collect", os.path.join(root, "scripts", "pmc_collect.py"))
    pmc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pmc)
    out, err = pmc.collect(["hbm", "valu"], 60.0, workload=["true"], per_pass=5.0)
    assert out is not None, err
    lin = out["kernels"]["k_lin_lm<3>"]
    assert lin["hbm_MB"] == (2 * 1000.0 + 500.0) * 1024 / 1e6 and abs(lin["hbm_bytes_per_observation"] - lin["hbm_MB"] * 1e6 / 10000) < 1e-9
    assert "valu_busy" not in lin and len(out["failed_passes"]) == 1 and "timed out" in out["failed_passes"][0]
    # bench.py: traffic from this run, valu_busy from the stored profile of the same sources
    sys.path.insert(0, root)
    import bench

    stored = {"kernel_source_sha16": bench.kernel_source_sha16(), "kernels": {"k_lin_lm<3>": {"valu_busy": 0.77, "hbm_bytes_per_observation": 1.0}}}
    pmc_file = tmp_path / "stored.json"
    pmc_file.write_text(json.dumps(stored))
    monkeypatch.setattr(bench, "PMC_FILE", str(pmc_file))
    roof = {"_lin_obs": 2.0e7, "_lin_s": 0.7e-3 * 250, "launches": 250, "algorithmic_bytes_per_launch": 1.9e9}
    schur = {}
    bench.fill_traffic(roof, schur, measure=True, timeout=15.0)
    assert roof["traffic_bytes_per_observation"] == lin["hbm_bytes_per_observation"] and "measured in this run" in roof["traffic_source"]
    assert roof["limited_by"]["valu_busy"] == 0.77 and "stored profile" in roof["traffic_source"] and "timed out" in roof["traffic_measurement_partial"]
