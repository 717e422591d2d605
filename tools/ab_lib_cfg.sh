#!/bin/bash
# A/B of two BUILDS of libcnn_amd.so on a stack config: tools/ab_lib_cfg.sh <config> <rounds> <alt.so>
CFG=$1; R=$2; ALT=$3
L=cnn_amd/lib/libcnn_amd.so
cp $L /tmp/new.so
run() { python bench.py --config $CFG --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for i in $(seq $R); do
  cp $ALT $L; echo "$CFG round $i [base] $(run)"
  cp /tmp/new.so $L; echo "$CFG round $i [new ] $(run)"
done
cp /tmp/new.so $L
