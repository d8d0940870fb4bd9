#!/bin/bash
# per-step cost of the panel factorisation in qrb_udt_kernel: kernel time with the 32 steps of every panel run 1 + N times
# (library built with make XFLAGS=-DQRB_X_EXTRA, copied to montecarlo.jl_amd/libdqmc_hip_extra.so; results are garbage for N > 0)
cd /tmp && export TMPDIR=/tmp
LOG=$GRAFT_REPO_ROOT/gpurun_out/qrb_time.log
: > $LOG
for N in 0 1 2; do
  rm -rf /tmp/qt_$N
  DQMC_HIP_LIB=$GRAFT_REPO_ROOT/montecarlo.jl_amd/libdqmc_hip_extra.so DQMC_QR_FORCE_TIMEOUT=extra:$N timeout -k 5 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qt_$N -o t -- python3 $GRAFT_REPO_ROOT/tools/qrb_time.py > /tmp/qt_$N.out 2>&1 || { echo "run $N failed" | tee -a $LOG; tail -5 /tmp/qt_$N.out | tee -a $LOG; exit 1; }
  f=$(find /tmp/qt_$N -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then echo "extra=$N: $(python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'qrb_udt' in r['Name']: print('calls', r['Calls'], 'avg ns', r['AverageNs'], 'min', r['MinNs'], 'max', r['MaxNs'])
")" | tee -a $LOG; else echo "extra=$N: no stats file" | tee -a $LOG; ls -R /tmp/qt_$N | head -20 | tee -a $LOG; fi
done
