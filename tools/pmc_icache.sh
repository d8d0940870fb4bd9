#!/bin/bash
# instruction-cache counters per kernel (ad hoc): rocprofv3 --pmc over the small fixed workloads, summary as text
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_icache
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|IFETCH|SQ_WAIT_INST|SQ_INST_LEVEL" | sort -u | head -40 > $OUT/avail.txt
echo "[pmc_icache] $(date +%T) pass 1"
rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/p1 -- python3 $R/tools/pmc_lds.py > /dev/null 2> $OUT/p1.err || echo "pass 1 failed"
echo "[pmc_icache] $(date +%T) pass 2"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/p2 -- python3 $R/tools/pmc_lds.py > /dev/null 2> $OUT/p2.err || echo "pass 2 failed"
cd $R
python3 - <<'PY' > $OUT/summary.txt
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc_icache"
acc = collections.defaultdict(list)
for f in glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void dqmc::", "").replace("dqmc::", "")[:40]
        acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
ks = sorted({k for k, _ in acc})
cs = sorted({c for _, c in acc})
for k in ks:
    print(k, {c: round(sum(acc[(k, c)]) / len(acc[(k, c)])) for c in cs if (k, c) in acc}, "launches", len(acc[(k, cs[0])]) if (k, cs[0]) in acc else 0)
PY
cat $OUT/avail.txt | head -20; cat $OUT/summary.txt
