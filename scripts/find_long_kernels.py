import csv,sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0].split("<")[0]
    rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),n))
rows.sort()
t0=rows[0][0]
for i,(s,e,n) in enumerate(rows):
    if n.startswith("k_db_") and e-s>200000:
        print("%8.2f ms  %-18s %8.1f us   prev: %s | next: %s"%((s-t0)/1e6,n,(e-s)/1e3, rows[i-1][2], rows[i+1][2] if i+1<len(rows) else ""))
# stage markers
for i,(s,e,n) in enumerate(rows):
    if n in ("k_pool_gram","k_sh_query","k_object_views","k_slots","k_concat"):
        print("%8.2f ms  %-18s %8.1f us"%((s-t0)/1e6,n,(e-s)/1e3))
