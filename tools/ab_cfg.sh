#!/bin/bash
# A/B of switches on a stack config: tools/ab_cfg.sh <config> <rounds> "<env A>" "<env B>" ...
CFG=$1; R=$2; shift; shift
for i in $(seq $R); do
  for V in "$@"; do
    E=""; [ "$V" != "-" ] && E="$V"
    val=$(env $E python bench.py --config $CFG --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'].split('|')[0], d['roofline']['frac'])")
    echo "$CFG round $i [$V] $val"
  done
done
