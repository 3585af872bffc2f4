#!/bin/bash
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/one.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from limo_amd import ba, default_options, synth
B=int(sys.argv[1]); ctx=ba.Context(0); o=default_options()
b=ba.Batch(ctx,[synth.make_window(5000+i) for i in range(B)]); b.solve(o)
PY
rocprofv3 --kernel-trace -d gpurun_out/prof_iter -o it$1 -- python /tmp/one.py $1 > gpurun_out/prof_run.log 2>&1
python scripts/iter_profile.py gpurun_out/prof_iter/it$1_results.db 130
