#!/bin/bash
# A/B of two BUILDS of libcnn_amd.so on one box: tools/ab_lib.sh <rounds> <alt.so>   (the in-tree build is "new", alt.so is "base")
R=$1; ALT=$2
L=cnn_amd/lib/libcnn_amd.so
cp $L /tmp/new.so
run() { python bench.py --no-extra-legs --no-stacks --no-conv-ns --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'].split('|')[0], d['roofline']['avg_us'])"; }
for i in $(seq $R); do
  cp $ALT $L; echo "round $i [base] $(run)"
  cp /tmp/new.so $L; echo "round $i [new ] $(run)"
done
cp /tmp/new.so $L
