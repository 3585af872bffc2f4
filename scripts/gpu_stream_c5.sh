#!/bin/bash
# BASELINE.json configs[4] at full length: 4541-frame synthetic drive through the keyframe_bundle_adjustment shim on
# the GPU (sliding 5-keyframe window, keyframe + landmark selection as the KITTI launch wires them): fps and ATE.
# Most of the wall time is the test's own scene synthesis (tracklets of ~7000 landmarks per frame), which the
# "back end" figure excludes.
mkdir -p gpurun_out
FRAMES=${1:-4541}
LMS=${2:-$((FRAMES * 31))}
exe=$(python -c "import sys; sys.path.insert(0,'tests'); import emu_ffi; print(emu_ffi.build_stream_test(gpu=True))")
( time timeout 800 $exe $FRAMES $LMS long ) 2>&1 | grep -E "^stream|CHECK|checks|real" | tee gpurun_out/stream_c5.log
# the product's own driver (limo_amd/kba/stream_driver.hpp: depth assignment from the sweep in the loop) on the same length
app=$(python -c "import sys; sys.path.insert(0,'tests'); import emu_ffi; print(emu_ffi.build_stream_app(gpu=True))")
( time timeout 1500 $app --frames $FRAMES --az ${AZ:-2000} --poses gpurun_out/limo_stream_poses.txt ) 2>&1 | grep -E "^limo_stream|^frame [0-9]*00:|real|^(fps|ate_rmse|ate_max|depth_fraction|keyframes|solves) " | tee gpurun_out/limo_stream_c5.log
