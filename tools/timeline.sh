#!/bin/bash
# one steady-state step of the default bench leg under rocprofv3 --kernel-trace -> gpurun_out/<tag>_timeline.txt
# usage (GPU box, repo root): bash tools/timeline.sh <tag> [extra bench.py args / env via "env A=1"]
TAG=${1:-tl}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl_$TAG
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $OUT/trace -o step --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-conv-ns --no-extra-legs --no-stacks "$@" > $OUT/bench.json 2> $OUT/trace.log)
python $GRAFT_REPO_ROOT/tools/step_timeline.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) 3 start:conv_fwd_pool_pk > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_timeline.txt
find $OUT -name "*kernel_trace.csv" -delete
cat $GRAFT_REPO_ROOT/gpurun_out/${TAG}_timeline.txt
