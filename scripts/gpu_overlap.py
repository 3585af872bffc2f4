"""How much do the slot groups' kernels overlap?  From a rocprofv3 kernel trace of the bench command: in a window of steady-state
rounds, the wall time, the time at least one / two / three kernels run, the summed kernel time per kernel name (stretched by the
overlap) - to hold against the one-group table.
   usage: python scripts/gpu_overlap.py bench_results.db"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tables else [t for t in tables if t.startswith("kernels")][0]
rows = list(db.execute("select name, start, end from %s order by start" % view))
rows = [(n.split("(")[0].replace("void ", "").replace("kba::", ""), s, e) for n, s, e in rows]
lin = [r for r in rows if r[0].startswith("k_lin_lm")]
# steady-state window: the middle third of the k_lin_lm launches
a, b = lin[len(lin) // 3][1], lin[2 * len(lin) // 3][1]
sel = [(n, max(s, a), min(e, b)) for n, s, e in rows if e > a and s < b]
ev = []
for n, s, e in sel:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, hist = 0, a, defaultdict(float)
for t, d in ev:
    hist[depth] += t - last
    last = t
    depth += d
hist[depth] += b - last
wall = b - a
print("window %.1f ms; kernels running: " % (wall / 1e6) + ", ".join("%d: %.1f %%" % (k, 100 * v / wall) for k, v in sorted(hist.items())))
per = defaultdict(lambda: [0, 0.0])
for n, s, e in sel:
    per[n][0] += 1
    per[n][1] += e - s
tot = sum(v[1] for v in per.values())
print("summed kernel time %.1f ms = %.2f x the wall time" % (tot / 1e6, tot / wall))
n_lin = per[[k for k in per if k.startswith("k_lin_lm")][0]][0]
for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:14]:
    print("   %-28s %6d launches  avg %8.1f us   %5.1f %% of the summed time" % (n[:28], c, t / c / 1e3, 100 * t / tot))
