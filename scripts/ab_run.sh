#!/bin/bash
# A/B of several prebuilt libraries inside ONE gpurun call (boxes differ by a few per cent, runs on one box by ~0.5 %):
#   bash scripts/ab_run.sh cfg4 A B C      -> per-kernel times of ab/A.so, ab/B.so, ab/C.so, twice each (ABC ABC)
# The library in place when the script ends is the LAST one named.
CFG=$1; shift
for rep in 1 2; do
  for n in "$@"; do
    cp ab/$n.so wayne_amd/libwayne_hip.so || exit 1
    echo -n "$n: "; python3 scripts/ab_kernels.py $CFG 40 || exit 1
  done
done
