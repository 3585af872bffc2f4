#!/bin/bash
# Where the time of the one-launch single-window solve goes (debug builds of the library; the product build is restored
# afterwards): -DKBA_COOP_TICKS = phase ticks (100 MHz) of workgroups 0, 1 and G - 1 of window 0; -DKBA_PROFILE_TICKS = the
# shader-clock stamps inside cam_assemble / cam_solve, of the one-launch kernel and of the stand-alone kernels.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from limo_amd import ba, default_options, synth
ctx = ba.Context(0); o = default_options()
for i in range(3):
    r = ctx.solve(synth.make_window(3002), o)
print("iterations", r["iterations_total"], "linearizations", r["num_linearizations"], "solves", r["num_solves"])
PY
if [ -z "$SKIP_COOP_TICKS" ]; then
LIMO_HIPCC_EXTRA="-DKBA_COOP_TICKS" python -c "import __graft_entry__ as g; g.build_hip(force=True)" 2>&1 | grep -i error
timeout 120 python /tmp/one.py 2>&1 | tail -4
for g in ${GS:-}; do echo "== G=$g"; KBA_COOP_G=$g timeout 120 python /tmp/one.py 2>&1 | tail -4; done
fi
LIMO_HIPCC_EXTRA="-DKBA_PROFILE_TICKS" python -c "import __graft_entry__ as g; g.build_hip(force=True)" 2>&1 | grep -i error
echo "== one launch"; timeout 120 python /tmp/one.py 2>&1 | grep "ticks" | tail -4
echo "== lock-step launches"; KBA_NO_COOP_SOLVE=1 timeout 120 python /tmp/one.py 2>&1 | grep "ticks" | tail -4
python -c "import __graft_entry__ as g; g.build_hip(force=True)"
