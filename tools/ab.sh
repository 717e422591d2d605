#!/bin/bash
# A/B of measurement switches on ONE box: tools/ab.sh <rounds> "<env A>" "<env B>" ...   ("-" = no switch)
# prints images/s of the default bench leg (C++ Layer API) for every variant, interleaved <rounds> times
R=$1; shift
for i in $(seq $R); do
  for V in "$@"; do
    E=""; [ "$V" != "-" ] && E="$V"
    val=$(env $E python bench.py --no-extra-legs --no-stacks --no-conv-ns --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "round $i [$V] $val"
  done
done
