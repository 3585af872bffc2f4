"""Virtual shards vs unsharded on the GPU for several P (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from limo_amd import ba, default_options, synth
ctx = ba.Context(0); o = default_options(); w = synth.config_c2()
wu = w.copy(); ru = ctx.solve(wu, o)
for P in [int(a) for a in sys.argv[1:]] or [2, 4, 5, 6, 7, 8]:
    for rep in range(2):
        ws = w.copy(); rs = ctx.solve_sharded(ws, o, P)
        print(P, rep, rs["final_cost"], abs(rs["final_cost"] - ru["final_cost"]) / ru["final_cost"], rs["iterations_total"], ru["iterations_total"], rs["termination"], rs["num_solves"], rs["n_trimmed_landmarks"], ru["n_trimmed_landmarks"], flush=True)
