import csv, glob, sys, collections
rows = collections.defaultdict(dict)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_step_dense64_f64" in r["Kernel_Name"]:
            d = rows[int(r["Dispatch_Id"])]
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(rows)
def mean(sel, name): 
    v = [rows[i].get(name, 0.0) for i in sel]
    return sum(v) / max(1, len(v))
one, fused = ids[10:30], ids[30:]
names = sorted({k for i in ids for k in rows[i]})
print("%-24s %16s %16s" % ("counter (mean per launch)", "one sweep", "10 fused sweeps"))
for n in names:
    print("%-24s %16.4g %16.4g" % (n, mean(one, n), mean(fused, n)))
