#!/bin/bash
# rocprofv3 capture of the default bench.py run (kernel trace + stats), then HBM-traffic PMC passes (separate runs).
# usage (on the GPU box, from the repo root): bash tools/profile_bench.sh <tag>
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch --output-format csv -- $BENCH --no-conv-ns > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o write --output-format csv -- $BENCH --no-conv-ns > /dev/null 2> $OUT/pmc_write.log
ls -R $OUT | head -30
