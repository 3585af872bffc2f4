#!/bin/bash
# Where the host time of the drive's C-ABI calls goes: KBA_HOST_TRACE (create / solve / download / destroy of limo_ba_solve and
# limo_ba_adjust_pose_only) and LIMO_SHIM_TRACE (the shim around them) over the first frames of the config-5 drive, averaged.
mkdir -p gpurun_out
app=$(python -c "import sys; sys.path.insert(0,'tests'); import emu_ffi; print(emu_ffi.build_stream_app(gpu=True))")
KBA_HOST_TRACE=1 LIMO_SHIM_TRACE=1 timeout 600 $app --frames ${1:-400} --az 2000 --quiet > gpurun_out/drive_trace_raw.txt 2>&1
python - <<'PY' | tee gpurun_out/drive_trace.txt
import re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open("gpurun_out/drive_trace_raw.txt"):
    l = l.strip()
    if l.startswith("[kba] ") or l.startswith("[shim] "):
        key = l.split(":")[0]
        rest = l.split(":", 1)[1] if ":" in l else l
        for m in re.finditer(r"([A-Za-z_+\- ]+?) (\d+(?:\.\d+)?) (us|ms)", rest):
            acc[key][m.group(1).strip(" ,")].append(float(m.group(2)))
        for m in re.finditer(r"(\d+) (LM iterations|solves|selected|observations)", l):
            acc[key]["# " + m.group(2)].append(float(m.group(1)))
for key, d in acc.items():
    if "window cut" in key: continue
    print(key)
    for name, v in d.items():
        v = v[len(v)//3:]
        print("   %-40s %9.1f (n=%d)" % (name, sum(v)/len(v), len(v)))
PY
grep -E "^limo_stream: (pipeline|host)" gpurun_out/drive_trace_raw.txt
