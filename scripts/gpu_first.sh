#!/bin/bash
# first GPU bring-up: ABI, parity tests, quick timing
set -x
mkdir -p gpurun_out
rocminfo | grep -E "gfx|Compute Unit" | head -4
nproc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -20
python -m pytest tests/test_gpu_ba.py -m gpu -x -q 2>&1 | tail -30
