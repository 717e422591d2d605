cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_first
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/f -o f --output-format csv -- python $GRAFT_REPO_ROOT/tools/tune_first_wgrad.py 256 2 > /dev/null 2> $OUT/log
python - <<'PY'
import csv,glob,collections,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_first/f/*counter_collection.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    if 'win' in n or 'pool_pk' in n: agg[n[:70]].append(float(r['Counter_Value']))
for k,v in agg.items(): print(k, round(2*1024*sum(v)/len(v)/1e6,1),'MB fetch', len(v))
PY
