#!/bin/bash
# the six 3x3 / stride-2 geometries of the BASELINE workloads, isolated: conv_rows_s2.hip against the kernels it replaces (CNN_AMD_CONV_S2=0)
cd "$(dirname "$0")/.."
OPS=${TUNE_OPS:-fwd,dgrad,dgrad_relu}
for geo in "256 16 55 55 32 3 2 0" "256 32 27 27 64 3 2 0" "256 64 13 13 128 3 2 0" "64 64 56 56 128 3 2 1" "64 128 28 28 256 3 2 1" "64 256 14 14 512 3 2 1"; do
  echo "== $geo"
  CNN_AMD_CONV_S2=2 TUNE_OPS=$OPS TUNE_NO_AUTOTUNE=1 python tools/one_layer.py $geo 20 2>&1 | grep -v rows_prep
  echo "  -- CONV_S2=0"
  CNN_AMD_CONV_S2=0 TUNE_OPS=$OPS TUNE_NO_AUTOTUNE=1 python tools/one_layer.py $geo 20 2>&1 | grep -v "rows_prep\|rd_prepare\|prepare"
done
