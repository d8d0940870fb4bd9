import csv, glob, json, sys, collections
out = collections.defaultdict(dict)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            if "sweep_chunk" in k or "qr_coop" in k:
                out[k][c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
print(json.dumps(out, indent=1))
