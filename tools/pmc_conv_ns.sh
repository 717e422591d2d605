#!/bin/bash
# HBM traffic of the three north-star convolution passes (3x3, 64 -> 128, 112x112, batch 256): FETCH_SIZE and WRITE_SIZE in separate
# rocprofv3 passes over tools/one_layer.py (MI355X_MICROARCH.md, HBM section), per launch, next to the algorithmic bytes.
# usage (GPU box, repo root): bash tools/pmc_conv_ns.sh <tag>   -> gpurun_out/prof_<tag>/hbm_traffic_conv_ns.json
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG/conv_ns
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/one_layer.py 256 64 112 112 128 3 1 0 2"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch --output-format csv -- $CMD > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o write --output-format csv -- $CMD > /dev/null 2> $OUT/pmc_write.log
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py $OUT $OUT/../hbm_traffic_conv_ns.json
find $OUT -name "*.csv" -size +4M -delete
python - <<PY
import json
j = json.load(open("$OUT/../hbm_traffic_conv_ns.json"))
x, y = 256 * 64 * 112 * 112 * 4, 256 * 128 * 110 * 110 * 4
print("algorithmic bytes: x %.1f MB + y %.1f MB = %.1f MB per pass" % (x / 1e6, y / 1e6, (x + y) / 1e6))
for k, v in j["kernels"].items():
    if v["hbm_bytes"] > 50e6:
        print("%-60s fetch %8.1f MB  write %8.1f MB  total %8.1f MB  (%.2f x algorithmic)" % (k[:60], v["fetch_bytes"] / 1e6, v["write_bytes"] / 1e6, v["hbm_bytes"] / 1e6, v["hbm_bytes"] / (x + y)))
PY
