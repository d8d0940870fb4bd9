#!/bin/bash
# PMC passes over tools/pmc_flush.py (run on the GPU box from the repo root); per-kernel means to gpurun_out/pmc_flush.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_flush
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/pmc_flush.py > /dev/null 2> $OUT/trace.err || echo "trace failed"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_$C -- python3 $R/tools/pmc_flush.py > /dev/null 2> $OUT/pmc_$C.err || echo "pmc $C failed"
done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS SQ_WAVES -d $OUT/pmc_sq -- python3 $R/tools/pmc_flush.py > /dev/null 2> $OUT/pmc_sq.err || echo "pmc sq failed"
cd $R
python3 - <<'P' > $R/gpurun_out/pmc_flush.txt
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc_flush"
for f in glob.glob(root + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sweep" in r["Name"]:
            print("trace", r["Name"][:70], r["Calls"], r["AverageNs"])
acc = collections.defaultdict(list)
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "sweep" in n:
            acc[(n.split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(k[0], k[1], "mean %.4g over %d" % (sum(v) / len(v), len(v)))
P
cat $R/gpurun_out/pmc_flush.txt
