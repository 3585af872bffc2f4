import sys, os, time
sys.path.insert(0, os.getcwd())
from limo_amd import ba, default_options, synth
ctx = ba.Context(0); o = default_options()
ws = [synth.make_window(5000 + i) for i in range(64)]
b = ba.Batch(ctx, ws)
for _ in range(3):
    b.reset(); b.solve(o)
import torch
ts=[]
for _ in range(15):
    torch.cuda.synchronize(); t0=time.perf_counter(); b.reset(); b.solve(o); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
ts.sort()
print("%-10s batch 64: median %.3f ms -> %.0f windows/s" % (sys.argv[1], 1e3*ts[len(ts)//2], 64/ts[len(ts)//2]))
