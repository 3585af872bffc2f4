#!/bin/bash
# Cycle counts of the phases of k_cam_assemble / k_cam_solve as seen by the first and the last lane of window 0 (a debug
# build of the library with -DKBA_PROFILE_TICKS; the product build is restored afterwards).  SIZES="1 512".
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from limo_amd import ba, default_options, synth
B=int(sys.argv[1]); ctx=ba.Context(0); o=default_options(max_num_iterations=1, num_trim_rounds=0)
b=ba.Batch(ctx,[synth.make_window(5000+i) for i in range(B)])
for _ in range(3):
    b.reset(); b.solve(o)
PY
LIMO_HIPCC_EXTRA="-DKBA_PROFILE_TICKS" python -c "import __graft_entry__ as g; g.build_hip(force=True)"
for B in ${SIZES:-1 512}; do echo "== $B windows"; python /tmp/one.py $B 2>&1 | grep "^\[ticks" | tail -4; done
python -c "import __graft_entry__ as g; g.build_hip(force=True)"
