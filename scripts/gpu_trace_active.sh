#!/bin/bash
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from limo_amd import ba, default_options, synth
B=int(sys.argv[1]); ctx=ba.Context(0); o=default_options()
b=ba.Batch(ctx,[synth.make_window(5000+i) for i in range(B)]); b.solve(o)
PY
KBA_TRACE_ACTIVE=1 python /tmp/one.py $1 2>&1 | grep "kba" | awk '{printf "%s ", $5} END {print ""}'
