"""Development aid: distribution of per-launch durations of selected kernels in a rocprofv3 rocpd database."""
import sqlite3, sys, collections
import numpy as np
db = sqlite3.connect(sys.argv[1])
names = sys.argv[2:]
d = collections.defaultdict(list)
for n, s, e in db.execute("select name, start, end from kernels"):
    n = n.replace("(anonymous namespace)::", "").split("(")[0]
    d[n].append((e - s) / 1000.0)
for k in names or sorted(d):
    a = np.array(d[k])
    if len(a) < 5:
        continue
    print("%-16s n %5d sum %9.0f median %7.1f p90 %7.1f p99 %8.1f max %8.1f  top10 %8.0f top50 %8.0f" % (
        k, len(a), a.sum(), np.median(a), np.percentile(a, 90), np.percentile(a, 99), a.max(), np.sort(a)[-10:].sum(), np.sort(a)[-50:].sum()))
