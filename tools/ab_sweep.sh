#!/bin/bash
# A/B timing of alternative builds of the library: tools/ab/<name>.so are selected in turn through DQMC_HIP_LIB
# (montecarlo.jl_amd/_lib.py); the product library is never touched
for rep in 1 2; do
for f in tools/ab/*.so; do
  echo -n "$(basename $f) : "
  DQMC_HIP_LIB="$PWD/$f" timeout -k 10 100 python tools/time_sweep_spatial.py | tail -1
done
done
