#!/bin/bash
# The product's streaming driver on the 4541-frame synthetic drive (BASELINE.json configs[4]) alone: fps, ATE, pose rows.
mkdir -p gpurun_out
FRAMES=${1:-4541}
app=$(python -c "import sys; sys.path.insert(0,'tests'); import emu_ffi; print(emu_ffi.build_stream_app(gpu=True))")
( time timeout 1500 $app --frames $FRAMES --az ${AZ:-2000} --poses gpurun_out/limo_stream_poses.txt ) 2>&1 | grep -E "^limo_stream|^frame [0-9]*000:|real|^(fps|ate_rmse|ate_max|depth_fraction|keyframes|solves) " | tee gpurun_out/limo_stream_c5.log
