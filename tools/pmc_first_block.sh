#!/bin/bash
# Where do the first block's kernels wait?  rocprofv3 --pmc passes (one counter group per run, kernel trace only -- never combined with
# other trace domains) over tools/first_block_pmc.py, averaged per launch and kernel into gpurun_out/pmc_first_block/summary.json.
# usage (GPU box, repo root): bash tools/pmc_first_block.sh
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_first_block
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_HIT_sum TCC_MISS_sum" \
           "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" \
           "TA_TA_BUSY_sum TCP_TA_TCP_STATE_READ_sum GRBM_TA_BUSY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/g$i -o g --output-format csv -- python $GRAFT_REPO_ROOT/tools/first_block_pmc.py 256 6 > /dev/null 2> $OUT/g$i.log || echo "group $i failed: $grp"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, json, os
out = os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/pmc_first_block'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
def short(n):
    for k in ('conv_fwd_pool_pk', 'conv_dgrad_pool', 'conv_dgrad_pk', 'conv_wgrad_win', 'elementwise_kernel', 'copy'):
        if k in n: return 'copy (torch)' if k in ('elementwise_kernel', 'copy') else k
    return None
for f in glob.glob(out + '/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if k: agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob(out + '/g1/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        if k: dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
res = {}
for k, cs in agg.items():
    # (the first launch of each kernel is a cold one: dropped)
    res[k] = {c: sum(v[1:]) / max(1, len(v) - 1) for c, v in cs.items()}
    if dur[k]: res[k]['us_under_profiler'] = sum(dur[k][1:]) / max(1, len(dur[k]) - 1)
json.dump(res, open(out + '/summary.json', 'w'), indent=1, sort_keys=True)
for k, v in res.items():
    print(k)
    for c, x in sorted(v.items()): print('   %-34s %16.1f' % (c, x))
PY
find $OUT -name "*.csv" -size +4M -delete
