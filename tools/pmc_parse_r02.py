"""Summarise the rocprofv3 --pmc passes of tools/profile_r02.sh per kernel: mean counter value per launch."""
import csv, glob, json, sys, collections
root = sys.argv[1]
out = collections.defaultdict(dict)
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void dqmc::", "").replace("dqmc::", "")
        acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        if any(s in k for s in ("gemm", "slab_chain", "sweep_lu4", "sweep_fused", "sweep_flush", "qr_coop", "qr_tail", "trsm_rl", "cb_apply")):
            out[k][c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
print(json.dumps(out, indent=1, sort_keys=True))
