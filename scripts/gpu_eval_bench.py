import sys, os
sys.path.insert(0, os.getcwd())
import bench
from limo_amd import ba, default_options
ws = bench.generate_windows([7000 + i for i in range(1024)], 5, 2000)
ctx = ba.Context(0)
print(bench.bench_evaluate(ctx, ws, default_options()))
