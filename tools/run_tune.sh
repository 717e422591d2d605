#!/bin/bash
# Time the reference net's four convolution layers (fwd / dgrad / wgrad kernels, isolated launches, batch 256) on the GPU
# box:  gpurun -- 'bash tools/run_tune.sh'        (CNN_AMD_IGEMM_CFG / CNN_AMD_WGRAD_CFG force a tile config, see the
# planners in conv_igemm.hip / conv_wgrad.hip)
mkdir -p gpurun_out
{
for shp in "256 3 224 224 16 3 2 0" "256 16 55 55 32 3 2 0" "256 32 27 27 64 3 2 0" "256 64 13 13 128 3 2 0" "256 64 112 112 128 3 1 0"; do
  echo "== $shp"
  for op in fwd dgrad wgrad; do TUNE_OPS=$op python tools/tune_conv.py $shp 5 2>&1 | grep -E " us " ; done
done
} > gpurun_out/tune_layers.log 2>&1
cat gpurun_out/tune_layers.log
