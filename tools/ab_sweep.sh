#!/bin/bash
# A/B timing of alternative builds of the library: tools/ab/<name>.so are swapped in turn
set -e
cp montecarlo.jl_amd/libdqmc_hip.so /tmp/orig.so
for rep in 1 2; do
for f in tools/ab/*.so; do
  cp "$f" montecarlo.jl_amd/libdqmc_hip.so
  echo -n "$(basename $f) : "
  timeout -k 10 100 python tools/time_sweep_spatial.py | tail -1
done
done
cp /tmp/orig.so montecarlo.jl_amd/libdqmc_hip.so
