"""Development aid: the merge fold's critical path from a rocprofv3 kernel_trace.csv -- for every kernel of a fold step, by its
position in the step (name # occurrence), the mean duration and the mean GAP in front of it (its start minus the end of the fold's
previous kernel): launch gaps between dependent kernels, and the host turn-arounds behind the two publishes.
usage: fold_timeline.py <kernel_trace.csv> [first_step last_step]"""
import csv, sys, collections
FOLD = ("k_db_", "k_ov_", "k_scan_lookback", "k_upload16", "k_publish", "k_ix_", "k_f_", "k_concat")
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    if n.startswith(FOLD):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n))
rows.sort()
# a step = from the first kernel behind a k_publish that follows a k_db_compact, to that publish
steps, cur, seen_compact = [], [], False
for s, e, n in rows:
    cur.append((s, e, n))
    if n.startswith("k_db_compact"):
        seen_compact = True
    if n == "k_publish" and seen_compact:
        steps.append(cur)
        cur, seen_compact = [], False
lo = int(sys.argv[2]) if len(sys.argv) > 2 else len(steps) // 10
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(steps)
dur, gap, cnt, order = collections.defaultdict(float), collections.defaultdict(float), collections.Counter(), {}
wall = 0.0
prev_end = None
for st in steps[lo:hi]:
    occ = collections.Counter()
    for s, e, n in st:
        occ[n] += 1
        key = "%s#%d" % (n, occ[n])
        order.setdefault(key, len(order))
        dur[key] += (e - s) / 1e3
        if prev_end is not None:
            gap[key] += (s - prev_end) / 1e3
        cnt[key] += 1
        prev_end = e
    wall += 0
ns = hi - lo
if ns <= 0:
    sys.exit("no fold steps found")
span = (steps[hi - 1][-1][1] - steps[lo][0][0]) / 1e3
print("fold steps %d..%d: %.1f us per step wall (first kernel of the first step to the last publish)" % (lo, hi, span / ns))
print("%-22s %6s %9s %9s" % ("kernel#occurrence", "calls", "dur us", "gap us"))
td = tg = 0.0
for key in sorted(order, key=order.get):
    c = cnt[key]
    print("%-22s %6d %9.2f %9.2f" % (key, c, dur[key] / c, gap[key] / c))
    td += dur[key] / ns
    tg += gap[key] / ns
print("per step: kernels %.1f us + gaps %.1f us = %.1f us" % (td, tg, td + tg))
