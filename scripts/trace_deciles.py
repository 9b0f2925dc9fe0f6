"""Development aid: per-kernel mean duration by call-index decile from a rocprofv3 kernel_trace.csv
(does a kernel's time follow the growing clouds of the fold, or is it a flat latency floor?)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    by[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if len(d) < 500:
        continue
    k = len(d) // 10
    print("%-28s %6d %8.1f ms |" % (n[:28], len(d), sum(d) / 1e3), " ".join("%6.1f" % (sum(d[i * k:(i + 1) * k]) / k) for i in range(10)))
