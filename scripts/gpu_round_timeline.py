"""Timeline of the streaming solve's LAST rounds from a rocprofv3 kernel trace: per kernel of a late round (few windows in flight) its
duration and the gap to the previous kernel on the same stream.
   usage (on the GPU box):  cd /tmp && export TMPDIR=/tmp; rocprofv3 --kernel-trace -d OUT -o t -- python scripts/gpu_sched_trace.py 1024
                            python scripts/gpu_round_timeline.py OUT/t_results.db [round-from-end]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 12
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tables if t.startswith("kernels") or t == "kernels"]
view = "kernels" if "kernels" in tables else kd[0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % view)]
rows = list(db.execute("select name, start, end, %s from %s order by start" % ("stream_id" if "stream_id" in cols else "queue_id", view)))
# rounds of ONE stream: split at k_sched_advance
by_stream = defaultdict(list)
for name, s, e, q in rows:
    by_stream[q].append((name.split("(")[0].replace("void ", "").replace("kba::", ""), s, e))
main = max(by_stream.values(), key=lambda v: sum(1 for n, _, _ in v if n.startswith("k_sched_advance")))
starts = [i for i, (n, _, _) in enumerate(main) if n.startswith("k_sched_advance")]
print("%d rounds on the main stream" % len(starts))
for which in (len(starts) // 3, len(starts) - back):
    a, b = starts[which], starts[which + 1]
    t0 = main[a][1]
    print("round %d of %d: %.1f us from its first kernel to the next round's first" % (which, len(starts), (main[b][1] - t0) / 1e3))
    prev_end = None
    for n, s, e in main[a:b]:
        print("   %-28s start %8.1f us  duration %7.1f us  gap %6.1f us" % (n[:28], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
        prev_end = e
