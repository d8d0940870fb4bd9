"""Summarise the rocprofv3 --pmc passes of tools/profile_r03.sh: mean counter value per launch and kernel
(r03_pmc_summary.json) and the HBM-side traffic of the most frequent MFMA launch against its algorithmic bytes
(r03_pmc_gemm.json).  Usage: pmc_parse.py <prof dir> <commit> <summary.json> <gemm.json> [source hash]"""
import csv, glob, json, sys, collections
root, commit, f_sum, f_gemm = sys.argv[1:5]
source_hash = sys.argv[5] if len(sys.argv) > 5 else None   # dqmc_build_source_hash() of the profiled library
out = collections.defaultdict(dict)
KEEP = ("gemm", "slab_chain", "sweep_lu4", "sweep_fused", "sweep_flush", "qrb_udt", "qr_coop", "qr_tail", "trsm_rl", "cb_apply", "udt_finish")
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void dqmc::", "").replace("dqmc::", "")
        acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        if any(s in k for s in KEEP):
            out[k][c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
out["_commit"] = commit
out["_source_hash"] = source_hash
out["_source"] = ("rocprofv3 --kernel-trace --pmc <counters> (one pass per counter group, tools/profile_r0N.sh) over tools/pmc_gemm.py "
                  "and tools/pmc_lds.py; config 3 shape: 32 units of 256 x 256; FETCH_SIZE / WRITE_SIZE in KB")
json.dump(out, open(f_sum, "w"), indent=1, sort_keys=True)

n, units = 256, 32
g = {"commit": commit, "source_hash": source_hash,
     "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes, tools/profile_r0N.sh) over "
               "tools/pmc_gemm.py (prepare + 20 wrap_greens launches + 2 sweep_spatial); 32 units of 256x256",
     "correction": "counters are in KB.  gfx950: FETCH_SIZE reports half of the bytes of wide coalesced streaming reads of 16 B per "
                   "lane (MI355X_MICROARCH.md, HBM section) -> x2 for those; WRITE_SIZE exact.  slab_chain_kernel reads its A operand "
                   "with 8-byte-per-lane loads, a width the guide calls uncalibrated: raw and doubled figures are both given, the "
                   "doubled one is used as traffic (upper bound)"}
sl = out.get("slab_chain_kernel", {})
if "FETCH_SIZE" in sl and "WRITE_SIZE" in sl:
    fe, wr = sl["FETCH_SIZE"]["mean_per_launch"], sl["WRITE_SIZE"]["mean_per_launch"]
    g["FETCH_SIZE_KB_per_launch"], g["WRITE_SIZE_KB_per_launch"] = fe, wr
    g["traffic_bytes_per_launch_raw"] = (fe + wr) * 1024
    g["traffic_bytes_per_launch"] = (2 * fe + wr) * 1024
    g["algorithmic_bytes_per_launch"] = units * 2 * n * n * 8 + 2 * n * n * 8
    g["kernel"] = "slab_chain_kernel (wrap_greens launches dominate the sample: read G, write G' per unit + two shared constants)"
json.dump(g, open(f_gemm, "w"), indent=1)
