mkdir -p gpurun_out
{
echo "== pytest conv"; timeout 900 python -m pytest tests -m gpu -q -x -k "conv" 2>&1 | tail -3
echo "== conv1"; TUNE_OPS=fwd,dgrad python tools/tune_conv.py 256 3 224 224 16 3 2 0 | grep gemm_
echo "== conv1 dgrad generic ck16 (7)/ck4?"; TUNE_OPS=dgrad CNN_AMD_IGEMM_CFG=8 python tools/tune_conv.py 256 3 224 224 16 3 2 0 | grep gemm_k
for L in "16 55 55 32" "32 27 27 64" "64 13 13 128"; do
  echo "== layer $L default"; TUNE_OPS=fwd,dgrad python tools/tune_conv.py 256 $L 3 2 0 | grep gemm_
done
} > gpurun_out/tune13.log 2>&1
grep -v amdgpu.ids gpurun_out/tune13.log
