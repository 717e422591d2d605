mkdir -p gpurun_out
{
echo "== pytest all"; timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for c in 200 208 207; do echo "== pytest conv cfg $c"; CNN_AMD_IGEMM_CFG=$c timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "mfma_vs_oracle and not k5 and not 224" 2>&1 | tail -3; done
echo "== NS default"; TUNE_OPS=fwd,dgrad python tools/tune_conv.py 256 64 112 112 128 3 1 0 | grep gemm_
echo "== bench"; timeout 600 python bench.py --steps 10 --warmup 3 --breakdown --no-cpu-baseline --no-conv-ns 2>&1 | grep -v amdgpu.ids | head -20
} > gpurun_out/tune9.log 2>&1
grep -v amdgpu.ids gpurun_out/tune9.log
