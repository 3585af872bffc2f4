"""Debugging aid: run with KBA_POISON=1 (every device block starts as NaN bytes).  Any result that differs from the
unpoisoned run means a kernel reads a word nobody wrote."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from limo_amd import ba, default_options, synth

ctx = ba.Context(0)
o = default_options()


def check(name, ws):
    outs = [w.copy() for w in ws]
    if len(ws) == 1:
        reps = [ctx.solve(outs[0], o)]
    else:
        b = ba.Batch(ctx, outs)
        b.solve(o)
        reps = b.download()
        outs = b.windows
        b.close()
    same = sum(np.array_equal(a.kf_pose, b_.kf_pose) for a, b_ in zip(ws, outs))
    nan = sum(not np.isfinite(b_.kf_pose).all() for b_ in outs)
    h = float(sum(np.abs(b_.kf_pose).sum() for b_ in outs))
    print("%-28s windows %4d unchanged %d nan %d terminations %s checksum %.12f" % (name, len(ws), same, nan, sorted(set(r["termination"] for r in reps)), h), flush=True)


short = len(sys.argv) > 1 and sys.argv[1] == "short"  # (three cases: the streaming batch, a single window, the large window)
ws = [synth.make_window(7000 + i) for i in range(64 if short else 300)]
check("single C2", ws[:1])
if not short:
    check("batch 8", ws[:8])
check("batch 64", ws[:64])
if not short:
    check("batch 300", ws)
check("single C4", [synth.config_c4()])
if not short:
    check("batch 40 + C4", ws[:40] + [synth.config_c4()])
