#!/bin/bash
# the 3x3 / stride-2 weight gradients of the BASELINE workloads, isolated: conv_wgrad_sp2.hip against the kernels it replaces (CNN_AMD_WGRAD_SP2=0)
cd "$(dirname "$0")/.."
for geo in "64 64 56 56 128 3 2 1" "64 128 28 28 256 3 2 1" "64 256 14 14 512 3 2 1" "256 32 27 27 64 3 2 0" "256 64 13 13 128 3 2 0"; do
  echo "== $geo"
  CNN_AMD_WGRAD_SP2=2 TUNE_OPS=wgrad TUNE_NO_AUTOTUNE=1 python tools/one_layer.py $geo 20 2>&1 | grep -v "amdgpu.ids"
  echo "  -- WGRAD_SP2=0"
  CNN_AMD_WGRAD_SP2=0 TUNE_OPS=wgrad TUNE_NO_AUTOTUNE=1 python tools/one_layer.py $geo 20 2>&1 | grep -v "amdgpu.ids"
done
