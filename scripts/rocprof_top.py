"""Development aid: per-kernel totals of a rocprofv3 run (rocpd sqlite output)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
print("%-44s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit %d" % n):
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    print("%-44s %8d %12.1f %10.2f %6.2f" % (name[:44], calls, total, avg, pct))
