#!/bin/bash
# limo_depth_estimate_begin/_end and the drive with the depth assignment one frame ahead: the parity test, then the 4541-frame
# drive with and without the prefetch (fps, pose rows compared).
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_depth.py -q -m gpu -k "two_halves or batch_equals" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_kba_shim.py -q -m gpu -k "emulated_drive" 2>&1 | tail -3
FRAMES=${1:-4541}
app=$(python -c "import sys; sys.path.insert(0,'tests'); import emu_ffi; print(emu_ffi.build_stream_app(gpu=True))")
for mode in thread stream none thread; do
  echo "== limo_stream --depth-ahead $mode"
  timeout 900 $app --frames $FRAMES --az 2000 --depth-ahead $mode --poses gpurun_out/poses_ahead_$mode.txt 2>&1 | grep -E "^limo_stream: (pipeline|host|depth)|^fps"
done | tee gpurun_out/prefetch_ab.log
md5sum gpurun_out/poses_ahead_*.txt | tee -a gpurun_out/prefetch_ab.log
