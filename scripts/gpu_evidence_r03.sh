#!/bin/bash
# Round-3 profiles in one gpurun call (everything lands in gpurun_out/r03/, the kept files are copied to profiles/r03_*):
#   1. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters of FULL rounds of the streaming solve, stamped
#      with the sha of the kernel sources they were taken on (bench.py only uses a profile of the sources it runs);
#   2. depth path: kernel tables (single frame, 32-frame batch), host-inclusive / device-resident rates, PMC traffic;
#   3. one C2 window per limo_ba_solve call and adjustPoseOnly: latency + kernel tables.
# Counter passes carry --kernel-trace only (no other trace domains).
OUT=gpurun_out/r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --kernel-trace -d $OUT/pmc -o p$i -- python scripts/pmc_round.py > $OUT/pmc_$i.log 2>&1 || echo "pass $i ($grp) failed: $(tail -2 $OUT/pmc_$i.log)"
done
python - <<'PY'
import sqlite3, glob, json, re, sys, os
sys.path.insert(0, os.getcwd())
import bench
OUT = "gpurun_out/r03"
meta = None
for f in sorted(glob.glob(OUT + "/pmc_*.log")):
    for line in open(f):
        if line.startswith("{"):
            meta = json.loads(line)
out = {"command": "scripts/gpu_evidence_r03.sh: rocprofv3 --pmc <group> --kernel-trace -- python scripts/pmc_round.py (1024 C2 windows, 1024 slots, one slot group; FULL rounds: per kernel the dispatch with the largest counter value / longest duration)",
       "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section: calibrated there for 16 B / lane loads; these kernels load 8 B / lane - the factor is kept because it reproduces the byte accounting of DESIGN.md 4, e.g. k_backsub 90 B counted vs 88 B accounted per observation); counters in KiB; WRITE_SIZE as is",
       "kernel_source_sha16": bench.kernel_source_sha16(), "batch": meta, "kernels": {}}
def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("kba::", "")
for db_path in sorted(glob.glob(OUT + "/pmc/*_results.db")):
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
        if meta and name.startswith(("k_lin_lm", "k_backsub")):
            k["hbm_bytes_per_observation"] = k["hbm_MB"] * 1e6 / meta["observations"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in k and k.get("SQ_BUSY_CYCLES"):
        k["mfma_busy_over_sq_busy"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / k["SQ_BUSY_CYCLES"]
    if meta and name.startswith("k_lin_lm") and "SQ_INSTS_VALU" in k:
        k["note_SQ_INSTS_VALU"] = "sampled on one shader engine: compare ratios, not absolute counts"
json.dump(out, open(OUT + "/pmc_kernels.json", "w"), indent=1)
for n, k in sorted(out["kernels"].items()):
    if n.startswith("k_"): print("%-34s %s" % (n[:34], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in k.items() if a in ("hbm_MB", "launch_us_under_counters", "hbm_TBps_under_counters", "hbm_bytes_per_observation", "mfma_busy_over_sq_busy")}))
PY
rm -rf $OUT/pmc
# ---- depth
./scripts/gpu_depth_prof.sh > $OUT/depth_prof.log 2>&1; PMC=1 DEPTH_ONLY_PMC=1 ./scripts/gpu_depth_prof.sh >> $OUT/depth_prof.log 2>&1
cp gpurun_out/rocprof_depth_single.txt gpurun_out/rocprof_depth_batch.txt gpurun_out/depth_rates.json gpurun_out/depth_pmc.json $OUT/ 2>/dev/null
tail -30 $OUT/depth_prof.log
# ---- one window at a time
./scripts/gpu_single_prof.sh > $OUT/single_window.txt 2>&1; head -20 $OUT/single_window.txt
./scripts/gpu_poseonly_prof.sh > $OUT/pose_only.txt 2>&1; head -14 $OUT/pose_only.txt
rm -rf gpurun_out/prof_single gpurun_out/prof_po gpurun_out/prof_depth gpurun_out/pmc_depth
