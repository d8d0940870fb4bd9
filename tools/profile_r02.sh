#!/bin/bash
# rocprofv3 summaries of round 2 (run on the GPU box from the repo root): kernel trace of the default bench,
# then PMC passes (separate runs, counters only) over the small fixed workloads of tools/pmc_*.py
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/trace.err || echo "trace run failed"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_$C -- python3 $R/tools/pmc_gemm.py > /dev/null 2> $OUT/pmc_$C.err || echo "pmc $C failed"
done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $OUT/pmc_sq -- python3 $R/tools/pmc_lds.py > /dev/null 2> $OUT/pmc_sq.err || echo "pmc sq failed"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT/pmc_sq2 -- python3 $R/tools/pmc_lds.py > /dev/null 2> $OUT/pmc_sq2.err || echo "pmc sq2 failed"
cd $R
python3 tools/pmc_parse_r02.py $OUT > $OUT/pmc_summary.json 2> $OUT/parse.err
find $OUT -name "*kernel_stats.csv" | head -3
ls $OUT
