#!/bin/bash
# Closing run of round 4 on the final sources: the bench line (in-run counter passes), then the drive with the depth thread and without.
OUT=gpurun_out/r04f
mkdir -p $OUT
( time python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real; tail -2 $OUT/bench.err; head -c 400 $OUT/bench.json; echo
app=$(python -c "import sys; sys.path.insert(0,'tests'); import emu_ffi; print(emu_ffi.build_stream_app(gpu=True))")
for mode in thread none thread; do
  echo "== limo_stream --depth-ahead $mode"
  timeout 900 $app --frames 4541 --az 2000 --depth-ahead $mode --poses $OUT/poses_$mode.txt 2>&1 | grep -E "^limo_stream: (pipeline|host|depth|ATE)|^fps"
done | tee $OUT/limo_stream_c5_gpu.log
md5sum $OUT/poses_*.txt | tee -a $OUT/limo_stream_c5_gpu.log
nproc; uptime
