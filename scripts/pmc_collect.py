"""Counter passes over FULL rounds of the streaming solve (scripts/pmc_round.py: 1024 C2 windows in 1024 slots, one slot group,
every window iterating), aggregated per kernel into one JSON file - the measurement behind `roofline.traffic` of bench.py
(which runs `--passes hbm` inside the bench run when rocprofv3 is on PATH) and behind profiles/rNN_pmc_kernels.json
(`--passes all`).

    python scripts/pmc_collect.py --out FILE [--passes hbm|all] [--timeout SEC]

Every pass is its own `rocprofv3 --pmc <group> --kernel-trace` run (FETCH_SIZE and WRITE_SIZE do not fit into one pass on
gfx950: MI355X_MICROARCH.md, "rocprofv3 PMC slots"; no other trace domain is combined with counters).  Units / corrections as
that guide's HBM section prescribes: counters in KiB, FETCH_SIZE x 2 on gfx950 (calibrated there for wide coalesced reads; these
kernels load 8 B per lane - the factor is kept because it reproduces the byte accounting of DESIGN.md 4), WRITE_SIZE as is.
Per kernel the dispatch with the largest counter value is taken (= a full round; the streaming solve also launches rounds in
which a kernel has nothing to do).  The file carries the sha of the kernel sources it was measured on."""
import argparse
import glob
import hashlib
import json
import os
import re
import shutil
import signal
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GROUPS = {
    "hbm": [["FETCH_SIZE"], ["WRITE_SIZE"]],
    "valu": [["SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT"]],
    "sq": [["SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"],
           ["GRBM_GUI_ACTIVE", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_INST_CYCLES_VMEM", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT"]],
}
SIMDS_PER_SAMPLED_SE = 32  # SQ counters are sampled on one shader engine (8 CUs x 4 SIMDs on this part): ratios, not absolutes


def kernel_source_sha16():
    h = hashlib.sha256()
    for name in ("kba_kernels.hip", "kba_items.hpp", "kba_math.hpp", "kba_layout.hpp", "kba_lm.hpp"):
        with open(os.path.join(ROOT, "limo_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").replace("kba::", "")


def _run_group(cmd, env, limit):
    """One rocprofv3 run in its own process group, so that a run that hangs is killed WITH the workload it started."""
    p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=limit)
        return p.returncode, out, err
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        p.communicate()
        return None, "", "timed out after %.0f s" % limit


def collect(passes, timeout, workload=None, per_pass=90.0):
    rocprof = shutil.which("rocprofv3")
    if not rocprof:
        return None, "rocprofv3 not on PATH"
    workload = workload or [sys.executable, os.path.join(ROOT, "scripts", "pmc_round.py")]
    groups = []
    for p in passes:
        groups += GROUPS[p]
    out = {"command": "scripts/pmc_collect.py: rocprofv3 --pmc <group> --kernel-trace -- python scripts/pmc_round.py (1024 C2 windows, 1024 slots, one slot group; "
                      "FULL rounds: per kernel the dispatch with the largest counter value / longest duration)",
           "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section); counters in KiB; WRITE_SIZE as is",
           "kernel_source_sha16": kernel_source_sha16(), "passes": [" ".join(g) for g in groups], "batch": None, "kernels": {}}
    # A pass that fails or hangs (seen once: a loaded box, the seven-counter SQ pass) is noted and SKIPPED - the passes before and
    # after it still count; every pass has its own time limit inside the overall one.
    t_end = time.time() + timeout
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("KBA_GROUPS", None)
    failed = []
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for i, grp in enumerate(groups):
            left = min(per_pass, t_end - time.time())
            if left < 3:
                failed.append("pass %d (%s): out of time" % (i, " ".join(grp)))
                continue
            cmd = [rocprof, "--pmc"] + grp + ["--kernel-trace", "-d", os.path.join(tmp, "p%d" % i), "-o", "p", "--"] + workload
            rc, r_out, r_err = _run_group(cmd, env, left)
            if rc is None:
                failed.append("pass %d (%s) %s" % (i, " ".join(grp), r_err))
                continue
            for line in r_out.splitlines():
                if line.startswith("{"):
                    out["batch"] = json.loads(line)
            dbs = glob.glob(os.path.join(tmp, "p%d" % i, "**", "*results.db"), recursive=True)
            if rc != 0 and not dbs:
                failed.append("pass %d (%s) failed: %s" % (i, " ".join(grp), (r_err or r_out)[-300:]))
                continue
            for db_path in dbs:
                db = sqlite3.connect(db_path)
                try:
                    rows = db.execute("select name, counter_name, max(counter_value), max(duration), count(*) from pmc_events group by name, counter_name").fetchall()
                except Exception as e:  # noqa: BLE001
                    failed.append("pass %d: %s" % (i, e))
                    continue
                for name, ctr, val, dur, _n in rows:
                    k = out["kernels"].setdefault(short(name), {})
                    k[ctr] = val
                    k["launch_us_under_counters"] = max(k.get("launch_us_under_counters", 0.0), dur / 1e3)
    if failed:
        out["failed_passes"] = failed
    if not out["kernels"]:
        return None, "; ".join(failed) or "no counter data"
    meta = out["batch"]
    for name, k in out["kernels"].items():
        if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            k["hbm_MB"] = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024 / 1e6
            k["hbm_TBps_under_counters"] = k["hbm_MB"] / k["launch_us_under_counters"] if k["launch_us_under_counters"] else None
            if meta:
                k["hbm_bytes_per_observation"] = k["hbm_MB"] * 1e6 / meta["observations"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in k and k.get("SQ_BUSY_CYCLES"):
            k["mfma_busy_over_sq_busy"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / k["SQ_BUSY_CYCLES"]
        if "SQ_ACTIVE_INST_VALU" in k and k.get("SQ_BUSY_CYCLES"):
            # quad-cycles the VALU port was issuing, over the busy cycles of the SIMDs of the sampled shader engine
            k["valu_busy"] = 4.0 * k["SQ_ACTIVE_INST_VALU"] / SIMDS_PER_SAMPLED_SE / k["SQ_BUSY_CYCLES"]
        if "SQ_LDS_BANK_CONFLICT" in k and k.get("SQ_INSTS_LDS"):
            k["lds_bank_conflict_per_lds_inst"] = k["SQ_LDS_BANK_CONFLICT"] / k["SQ_INSTS_LDS"]
    if meta:
        tot = sum(k.get("hbm_MB", 0.0) for n, k in out["kernels"].items() if n.startswith("k_"))
        if tot > 0.0:
            out["round_hbm_MB"] = tot
            out["round_hbm_bytes_per_observation"] = tot * 1e6 / meta["observations"]
    return out, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--passes", default="hbm", help="comma-separated: hbm, valu, sq; or all")
    ap.add_argument("--timeout", type=float, default=420.0, help="seconds for all passes together")
    ap.add_argument("--per-pass", type=float, default=90.0, help="seconds one pass may take (a pass that exceeds it is killed and skipped)")
    a = ap.parse_args()
    passes = ["hbm", "sq"] if a.passes == "all" else a.passes.split(",")
    out, err = collect(passes, a.timeout, per_pass=a.per_pass)
    if out is None:
        sys.stderr.write("pmc_collect: %s\n" % err)
        return 1
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    for n, k in sorted(out["kernels"].items()):
        if n.startswith("k_"):
            print("%-34s %s" % (n[:34], {x: (round(y, 3) if isinstance(y, float) else y) for x, y in k.items() if x in ("hbm_MB", "hbm_bytes_per_observation", "launch_us_under_counters", "valu_busy", "mfma_busy_over_sq_busy", "lds_bank_conflict_per_lds_inst")}))
    print("round: %.1f MB = %.1f B per observation" % (out.get("round_hbm_MB", 0.0), out.get("round_hbm_bytes_per_observation", 0.0)))
    for f in out.get("failed_passes", []):
        sys.stderr.write("pmc_collect: skipped %s\n" % f)
    return 0


if __name__ == "__main__":
    sys.exit(main())
