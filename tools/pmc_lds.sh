#!/bin/bash
# LDS bank-conflict counters of the kernels matching $1 in a short default bench run -> gpurun_out/pmc_lds_<tag>.txt
# usage (GPU box): bash tools/pmc_lds.sh <kernel regex> <tag> [bench args]
RE=$1; TAG=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_lds_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_]*LDS[A-Z_]*\|SQ_ACTIVE_INST_LDS\|SQ_INSTS_LDS\|SQ_WAIT_INST_LDS" | sort -u > $OUT/available.txt
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT"; do
  D=$OUT/$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "$RE" -d $D -o c --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-conv-ns --no-extra-legs --no-stacks "$@" > /dev/null 2> $D.log
done
python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/pmc_lds_$TAG.txt
import csv,glob,collections,os
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_lds_'+os.environ.get('TAG','')
for f in sorted(glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_lds_*/*/*counter_collection.csv')+glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_lds_*/*/*/*counter_collection.csv')):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:110], r["Counter_Name"])].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print(k[0], k[1], round(sum(v)/len(v),1), 'x', len(v))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_lds_$TAG.txt; cat $OUT/available.txt | tr '\n' ' '
