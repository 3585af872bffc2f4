#!/bin/bash
# Per-kernel HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters (MFMA / VALU / wait cycles) of FULL
# rounds of the streaming solve -> gpurun_out/r02_pmc_kernels.json (copy to profiles/).  Counter passes carry
# --kernel-trace only (no other trace domains).
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d gpurun_out/pmc_r02 -o p$i -- python scripts/pmc_round.py > gpurun_out/pmc_r02_$i.log 2>&1 || echo "pass $i ($grp) failed: $(tail -2 gpurun_out/pmc_r02_$i.log)"
done
python - <<'PY'
import sqlite3, glob, json, re
meta = None
for f in sorted(glob.glob("gpurun_out/pmc_r02_*.log")):
    for line in open(f):
        if line.startswith("{"):
            meta = json.loads(line)
out = {"command": "scripts/gpu_pmc_r02.sh: rocprofv3 --pmc <group> --kernel-trace -- python scripts/pmc_round.py (1024 C2 windows, 1024 slots, one slot group; FULL rounds: per kernel the dispatch with the largest counter value / longest duration)",
       "correction": "gfx950: FETCH_SIZE x2 for wide coalesced reads (MI355X_MICROARCH.md, HBM section); counters in KiB; WRITE_SIZE as is",
       "batch": meta, "kernels": {}}
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("kba::", "")
    return n
for db_path in sorted(glob.glob("gpurun_out/pmc_r02/*_results.db")):
    db = sqlite3.connect(db_path)
    try:
        rows = db.execute("select name, counter_name, max(counter_value), max(duration), count(*) from pmc_events group by name, counter_name").fetchall()
    except Exception as e:
        print(db_path, e); continue
    for name, ctr, val, dur, n in rows:
        k = out["kernels"].setdefault(short(name), {})
        k[ctr] = val
        k["launch_us_under_counters"] = max(k.get("launch_us_under_counters", 0.0), dur / 1e3)
for name, k in out["kernels"].items():
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        k["hbm_MB"] = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024 / 1e6
        k["hbm_TBps_under_counters"] = k["hbm_MB"] / k["launch_us_under_counters"] if k["launch_us_under_counters"] else None
        if meta and ("k_linearize" in name or "k_lin_lm" in name):
            k["hbm_bytes_per_observation"] = k["hbm_MB"] * 1e6 / meta["observations"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in k and k.get("SQ_BUSY_CYCLES"):
        k["mfma_busy_over_sq_busy"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / k["SQ_BUSY_CYCLES"]
lin = [n for n in out["kernels"] if n.startswith("k_lin_lm") or n.startswith("k_linearize")]
if lin:
    out["kernels"]["k_linearize"] = out["kernels"][lin[0]]
json.dump(out, open("gpurun_out/r02_pmc_kernels.json", "w"), indent=1)
for n, k in sorted(out["kernels"].items()):
    print("%-34s %s" % (n[:34], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in k.items()}))
PY
