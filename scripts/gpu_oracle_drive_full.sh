#!/bin/bash
# configs[4] at full length against the ORACLE: the 4541-frame drive through liblimo_hip.so and the same drive with the CPU oracle
# behind every C-ABI call (tests/cpp/oracle_abi.cpp), compared call by call (tests/stream_compare.py).  The oracle drive is CPU
# work (a few minutes on the box's host cores).   usage: scripts/gpu_oracle_drive_full.sh [frames]
FR=${1:-4541}
mkdir -p gpurun_out
gpu=$(python -c "import sys; sys.path.insert(0,'tests'); import emu_ffi; print(emu_ffi.build_stream_app(gpu=True))")
orc=$(python -c "import sys; sys.path.insert(0,'tests'); import emu_ffi; print(emu_ffi.build_stream_app(oracle=True))")
( time ORACLE_ABI_THREADS=8 LIMO_STREAM_TRACE=1 timeout 1500 $orc --frames $FR --az 2000 --quiet --poses gpurun_out/oracle_full_poses.txt 2> gpurun_out/oracle_full.trace | grep -E "^limo_stream: (pipeline|ATE)" ) 2>&1 | tail -4 &
LIMO_STREAM_TRACE=1 timeout 600 $gpu --frames $FR --az 2000 --quiet --poses gpurun_out/gpu_full_poses.txt 2> gpurun_out/gpu_full.trace | grep -E "^limo_stream: (pipeline|ATE)"
wait
python - <<PY
import sys, numpy as np
sys.path.insert(0, "tests")
import stream_compare as sc
r = sc.compare(open("gpurun_out/gpu_full.trace").read(), open("gpurun_out/oracle_full.trace").read())
a, b = np.loadtxt("gpurun_out/gpu_full_poses.txt"), np.loadtxt("gpurun_out/oracle_full_poses.txt")
d = np.linalg.norm(a[:, [3, 7, 11]] - b[:, [3, 7, 11]], axis=1)
print("GPU vs oracle drive, $FR frames:", r)
print("pose rows: largest position difference %.3e m (frame %d), rms %.3e m" % (d.max(), int(d.argmax()), float(np.sqrt((d * d).mean()))))
open("gpurun_out/oracle_drive_full.txt", "w").write(repr(r) + "\npose rows: max %.3e m at frame %d, rms %.3e m\n" % (d.max(), int(d.argmax()), float(np.sqrt((d * d).mean()))))
PY
rm -f gpurun_out/gpu_full.trace gpurun_out/oracle_full.trace
