#!/bin/bash
# 3x3 / stride-1 layers with plane widths outside the BASELINE instances (SURVEY 8(d)'s pad-0 VGG shapes), isolated, batch 128:
# conv_rows_any.hip against the kernels it replaces (CNN_AMD_ROWS_ANY=0)
cd "$(dirname "$0")/.."
OPS=${TUNE_OPS:-fwd,dgrad_relu,wgrad}
for geo in "128 64 222 222 64 3 1 0" "128 64 109 109 128 3 1 0" "128 128 52 52 256 3 1 0" "128 256 50 50 256 3 1 0" "128 256 23 23 512 3 1 0" "128 512 21 21 512 3 1 0" "64 64 96 96 128 3 1 1"; do
  echo "== $geo"
  TUNE_OPS=$OPS TUNE_NO_AUTOTUNE=1 python tools/one_layer.py $geo 5 2>&1 | grep -v "amdgpu.ids\|rows_prep"
  echo "  -- ROWS_ANY=0 WGRAD_SP_ANY=0"
  CNN_AMD_ROWS_ANY=0 CNN_AMD_WGRAD_SP_ANY=0 TUNE_OPS=$OPS python tools/one_layer.py $geo 5 2>&1 | grep -v "amdgpu.ids\|prep"
done
