#!/bin/bash
# Where the time of the one-launch adjustPoseOnly kernel (k_solve_wg) goes on the windows of the config-5 drive: debug build of
# the library with -DKBA_WG_TICKS (phase ticks of lane 0, 100 MHz), the drive for a few hundred frames, the phase sums averaged per
# call; the product build is restored afterwards.
mkdir -p gpurun_out
LIMO_HIPCC_EXTRA="-DKBA_WG_TICKS" python -c "import __graft_entry__ as g; g.build_hip(force=True)" 2>&1 | grep -i error
app=$(python -c "import sys; sys.path.insert(0,'tests'); import emu_ffi; print(emu_ffi.build_stream_app(gpu=True))")
timeout 600 $app --frames ${1:-300} --az 2000 --quiet > gpurun_out/wg_ticks_raw.txt 2>&1
grep -c "wg ticks" gpurun_out/wg_ticks_raw.txt
python - <<'PY' | tee gpurun_out/wg_ticks.txt
import re
rows = []
for l in open("gpurun_out/wg_ticks_raw.txt"):
    if "[wg ticks]" not in l: continue
    head, tail = l.split(":", 1)
    h = [int(x) for x in re.findall(r"(\d+) (?:obs|landmark blocks|iterations|linearisations)", head)]
    t = [int(x) for x in re.findall(r"(-?\d+)(?= \||\s*\(x10)", tail)]
    names = re.findall(r"([a-z+\- ]+?) -?\d+", tail)
    rows.append((h, t))
rows = rows[len(rows)//3:]
n = len(rows)
names = ["init", "view consts", "lin", "assemble", "decide-lin", "damp", "cam solve", "backsub", "reduce+decide", "accept", "trim"]
print("%d calls: %.0f obs, %.1f landmark blocks, %.1f iterations, %.1f linearisations per call" % (n, *[sum(r[0][k] for r in rows)/n for k in range(4)]))
tot = 0
for k, nm in enumerate(names):
    v = sum(r[1][k] for r in rows)/n/100.0
    tot += v
    print("  %-14s %7.1f us per call" % (nm, v))
print("  %-14s %7.1f us per call" % ("sum", tot))
PY
python -c "import __graft_entry__ as g; g.build_hip(force=True)" 2>&1 | grep -i error
