#!/bin/bash
# FETCH_SIZE per kernel of the headline step (one rocprofv3 --pmc pass) -> gpurun_out/pmc_step/fetch.txt; then an unprofiled bench line
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_step
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-conv-ns --no-layer-api > /dev/null 2> $OUT/log
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os, re
f = glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_step/pmc_fetch/*counter_collection.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = re.sub(r"\(anonymous namespace\)::", "", r['Kernel_Name'])
    n = re.sub(r"^void ", "", n).split('(')[0]
    agg[n].append(float(r['Counter_Value']))
tot = 0
lines = []
for k, v in agg.items():
    mb = 2 * 1024 * sum(v) / len(v) / 1e6
    if len(v) >= 20: tot += mb * (len(v) // 25 if len(v) >= 25 else 1)
    lines.append((mb, k, len(v)))
for mb, k, n in sorted(lines, reverse=True)[:16]: print(f"{mb:9.1f} MB fetch/launch  x{n:3d}  {k[:80]}")
PY
find $OUT -name "*.csv" -size +4M -delete
