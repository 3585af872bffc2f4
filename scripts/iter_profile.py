"""Per-LM-iteration device time from a rocprofv3 kernel trace (sqlite): iterations are delimited by k_lm_damp."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = list(db.execute("select name, start, end from kernels order by start"))
t0 = rows[0][1]
it = -1
acc = collections.OrderedDict()
per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
for name, s, e in rows:
    short = name.split("(")[0].split("::")[-1]
    if short == "k_solve_init":
        it = -1
    if short == "k_lm_damp":
        it += 1
    key = it
    acc.setdefault(key, [0.0, s, e])
    acc[key][0] += (e - s) / 1e3
    acc[key][2] = e
    per_kernel[key][short] += (e - s) / 1e3
print("iter  busy_us  wall_us   top kernels")
for k, (busy, s, e) in list(acc.items())[: int(sys.argv[2]) if len(sys.argv) > 2 else 200]:
    top = sorted(per_kernel[k].items(), key=lambda x: -x[1])[:4]
    print("%4d %8.1f %8.1f   %s" % (k, busy, (e - s) / 1e3, "  ".join("%s=%.0f" % t for t in top)))
