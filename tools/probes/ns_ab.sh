# north-star shape + two stack shapes, forward / data gradient on the implicit GEMM (tuner off, rule-based tiles)
for shape in "256 64 112 112 128 3 1 0" "128 256 56 56 256 3 1 1" "64 64 56 56 64 3 1 1" "64 128 28 28 128 3 1 1"; do
  echo "== $shape"
  env TUNE_OPS=fwd,dgrad CNN_AMD_IGEMM_AUTOTUNE=0 CNN_AMD_FWD_RD=0 CNN_AMD_DGRAD_RD=0 python tools/tune_conv.py $shape 5 2>/dev/null | grep -E "igemm" | grep -v prep
done
