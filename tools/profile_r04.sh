#!/bin/bash
# rocprofv3 summaries of round 4 (run on the GPU box from the repo root): kernel trace of the default bench, then PMC
# passes (separate runs, counters only) over the small fixed workloads of tools/pmc_*.py.  Everything judged is copied
# into gpurun_out/prof_r04/summary/ with the commit the library was built from (profiles/r04_commit.txt, written before
# the gpurun call - the box has no .git).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_r04
rm -rf $OUT; mkdir -p $OUT/summary
COMMIT=$(cat $R/profiles/r04_commit.txt 2>/dev/null || echo unknown)
HASH=$(python3 -c "import ctypes; L = ctypes.CDLL('$R/montecarlo.jl_amd/libdqmc_hip.so'); L.dqmc_build_source_hash.restype = ctypes.c_char_p; print(L.dqmc_build_source_hash().decode())" 2>/dev/null || echo unknown)
cd /tmp && export TMPDIR=/tmp
echo "[profile_r04] $(date +%T) next: rocprofv3 --kernel-trace --stats --output-format csv -d $OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/summary/r04_bench_under_rocprof.json 2> $OUT/trace.err || echo "trace run failed"
for C in FETCH_SIZE WRITE_SIZE; do
  echo "[profile_r04] $(date +%T) next: rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OU"
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_$C -- python3 $R/tools/pmc_gemm.py > /dev/null 2> $OUT/pmc_$C.err || echo "pmc $C failed"
  echo "[profile_r04] $(date +%T) next: rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OU"
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $OUT/pmc_qr_$C -- python3 $R/tools/pmc_lds.py > /dev/null 2> $OUT/pmc_qr_$C.err || echo "pmc qr $C failed"
done
echo "[profile_r04] $(date +%T) next: rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_C"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $OUT/pmc_sq -- python3 $R/tools/pmc_lds.py > /dev/null 2> $OUT/pmc_sq.err || echo "pmc sq failed"
echo "[profile_r04] $(date +%T) next: rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT/pmc_sq2 -- python3 $R/tools/pmc_lds.py > /dev/null 2> $OUT/pmc_sq2.err || echo "pmc sq2 failed"
echo "[profile_r04] $(date +%T) next: rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d $OUT/pmc_sq3 -- python3 $R/tools/pmc_gemm.py > /dev/null 2> $OUT/pmc_sq3.err || echo "pmc sq3 failed"
cd $R
python3 tools/pmc_parse.py $OUT "$COMMIT" $OUT/summary/r04_pmc_summary.json $OUT/summary/r04_pmc_gemm.json "$HASH" 2> $OUT/parse.err
# the bench line last: it reads the PMC summary just written (profiles/ on the box is this run's scratch copy)
cp $OUT/summary/r04_pmc_summary.json $OUT/summary/r04_pmc_gemm.json $R/profiles/
echo "[profile_r04] $(date +%T) next: python3 $R/bench.py > $OUT/summary/r04_bench.json"
python3 $R/bench.py > $OUT/summary/r04_bench.json 2> $OUT/bench.err || echo "bench failed"
ST=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
if [ -n "$ST" ]; then (echo "# rocprofv3 --kernel-trace --stats of: bench.py --steps 5 --warmup 1 --no-cpu-baseline; library built from commit $COMMIT, kernel source hash $HASH"; cat $ST) > $OUT/summary/r04_kernel_stats.csv; fi
ls $OUT/summary
