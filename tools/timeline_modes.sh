#!/bin/bash
# kernel timeline of one steady-state step of the headline leg under a few scheduling switches -> gpurun_out/tl_<mode>.txt
cd /tmp && export TMPDIR=/tmp
for mode in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/tl_$mode; rm -rf $OUT; mkdir -p $OUT
  unset CNN_AMD_NO_DEFER_DX0 CNN_AMD_NO_EARLY_UPDATE
  if [ $mode = nodefer ]; then export CNN_AMD_NO_DEFER_DX0=1; fi
  if [ $mode = noearly ]; then export CNN_AMD_NO_EARLY_UPDATE=1; fi
  rocprofv3 --kernel-trace -d $OUT -o step --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-conv-ns --no-layer-api > /dev/null 2> $OUT/log
  python $GRAFT_REPO_ROOT/tools/step_timeline.py $(find $OUT -name "*kernel_trace.csv" | head -1) 3 > $GRAFT_REPO_ROOT/gpurun_out/tl_$mode.txt
  find $OUT -name "*kernel_trace.csv" -delete
done
