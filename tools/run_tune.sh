mkdir -p gpurun_out
{
echo "== pytest all"; timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
echo "== conv1 direct"; TUNE_OPS=fwd,dgrad python tools/tune_conv.py 256 3 224 224 16 3 2 0 | grep -v prep
echo "== conv1 mfma (CNN_AMD_NO_DIRECT)"; CNN_AMD_NO_DIRECT=1 TUNE_OPS=fwd,dgrad python tools/tune_conv.py 256 3 224 224 16 3 2 0 | grep gemm_
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-conv-ns 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'])"
} > gpurun_out/tune21.log 2>&1
grep -v amdgpu.ids gpurun_out/tune21.log
