mkdir -p gpurun_out
{
echo "== pytest all"; timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3
echo "== NS wgrad"; TUNE_OPS=wgrad python tools/tune_conv.py 256 64 112 112 128 3 1 0 | grep -E "wgrad_k"
for L in "3 224 224 16" "16 55 55 32" "32 27 27 64" "64 13 13 128"; do
  echo "== layer $L"; TUNE_OPS=wgrad python tools/tune_conv.py 256 $L 3 2 0 | grep -E "wgrad_k"
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-conv-ns 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
} > gpurun_out/tune30.log 2>&1
grep -v amdgpu.ids gpurun_out/tune30.log
