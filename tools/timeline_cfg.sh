#!/bin/bash
# one steady-state step of a stack config under rocprofv3 --kernel-trace -> gpurun_out/<tag>_timeline.txt
# usage (GPU box, repo root): [ENV=..] bash tools/timeline_cfg.sh <tag> <config> <first kernel of a step, e.g. conv_stem_fwd>
TAG=$1; CFG=$2; FIRST=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/tl_$TAG
rm -rf $OUT; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace -d $OUT/trace -o step --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 6 --warmup 3 --no-cpu-baseline > $OUT/bench.json 2> $OUT/trace.log)
python $GRAFT_REPO_ROOT/tools/step_timeline.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) 2 start:$FIRST > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_timeline.txt
find $OUT -name "*kernel_trace.csv" -delete
