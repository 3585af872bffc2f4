"""Print the rocprofv3 --kernel-trace --stats summary (top kernels) from its sqlite output."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("%-64s %8s %12s %10s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
for name, calls, tot, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print("%-64s %8d %12.3f %10.2f %7.2f" % (name[:64], calls, tot / 1e3, avg, pct))
