"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, dispatch count and mean counter value."""
import csv, json, sys, collections
path, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
with open(path) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name", r.get("Kernel Name", "?")).split("(")[0]
        name = name.replace("void ", "").split("<")[0].strip()          # k_ov_query<false, false> -> k_ov_query
        a = acc[name][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
res = {k: {c: dict(dispatches=v[0], mean=v[1] / max(v[0], 1), total=v[1]) for c, v in d.items()} for k, d in acc.items()}
json.dump(res, open(out, "w"), indent=1)
print("kernels:", len(res))
