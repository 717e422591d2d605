#!/bin/bash
# like tools/ab.sh, but a variant is "<env assignments or -> | <extra bench.py args>"
R=$1; shift
for i in $(seq $R); do
  for V in "$@"; do
    E="${V%%|*}"; A="${V#*|}"; [ "$E" == "-" ] && E=""
    val=$(env $E python bench.py --no-extra-legs --no-stacks --no-conv-ns --no-cpu-baseline $A 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "round $i [$V] $val"
  done
done
