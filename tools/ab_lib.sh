#!/bin/bash
# A/B of two BUILDS on one box: tools/ab_lib.sh <rounds> <alt libcnn_amd.so> [<alt libcnn_amd_host.so>]   (in-tree = "new", alt = "base")
R=$1; ALT=$2; ALTH=$3
L=cnn_amd/lib/libcnn_amd.so; LH=cnn_amd/lib/libcnn_amd_host.so
cp $L /tmp/new.so; cp $LH /tmp/new_host.so
run() { python bench.py --no-extra-legs --no-stacks --no-conv-ns --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'].split('|')[0], d['roofline']['avg_us'])"; }
for i in $(seq $R); do
  cp $ALT $L; [ -n "$ALTH" ] && cp $ALTH $LH; echo "round $i [base] $(run)"
  cp /tmp/new.so $L; cp /tmp/new_host.so $LH; echo "round $i [new ] $(run)"
done
cp /tmp/new.so $L; cp /tmp/new_host.so $LH
