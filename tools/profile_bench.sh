#!/bin/bash
# rocprofv3 captures of bench.py for a round: kernel trace + stats of the three BASELINE workloads, then the HBM-traffic PMC
# passes of the headline configuration (separate runs, as MI355X_MICROARCH.md prescribes; never combined with other trace domains).
# usage (on the GPU box, from the repo root): bash tools/profile_bench.sh <tag>     -> gpurun_out/prof_<tag>/
TAG=${1:-r06}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
# (the default command also runs a few steps of configs[3] / [4]: the first trace therefore holds their kernels too)
rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- $BENCH > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch --output-format csv -- $BENCH --no-conv-ns --no-extra-legs --no-stacks > /dev/null 2> $OUT/pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o write --output-format csv -- $BENCH --no-conv-ns --no-extra-legs --no-stacks > /dev/null 2> $OUT/pmc_write.log
for CFG in vgg11 resnet18; do
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$CFG -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_${CFG}_under_rocprof.json 2> $OUT/trace_$CFG.log
done
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py $OUT $OUT/hbm_traffic.json
# timeline of one steady-state step of the HEADLINE leg (no layer-api / conv_ns legs in that process)
(cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace_step -o step --output-format csv -- $BENCH --no-conv-ns --no-extra-legs --no-stacks > /dev/null 2> $OUT/trace_step.log)
python tools/step_timeline.py $(find $OUT/trace_step -name "*kernel_trace.csv" | head -1) 3 > $OUT/step_timeline.txt
# the same step with a ONE-rank RCCL communicator and the exchange forced on: where do the two all-reduce kernels sit?
(cd /tmp && rocprofv3 --kernel-trace -d $OUT/trace_step_comm -o step --output-format csv -- $BENCH --no-conv-ns --no-extra-legs --no-stacks --one-rank-comm > $OUT/bench_one_rank_comm.json 2> $OUT/trace_step_comm.log)
python tools/step_timeline.py $(find $OUT/trace_step_comm -name "*kernel_trace.csv" | head -1) 3 start:conv_fwd_pool_pk > $OUT/step_timeline_one_rank_comm.txt
# plain (unprofiled) runs of the same commands: the numbers the profiles are read against
python bench.py --steps 100 --warmup 20 --staged-input --breakdown > $OUT/bench_alexnet.json 2> $OUT/bench_alexnet_breakdown.txt
python bench.py --config vgg11 --breakdown > $OUT/bench_vgg11.json 2> $OUT/bench_vgg11_breakdown.txt
python bench.py --config resnet18 --breakdown > $OUT/bench_resnet18.json 2> $OUT/bench_resnet18_breakdown.txt
# the ResNet-shaped step with every backward pass serial (no weight gradient beside the data gradient): each kernel's duration ALONE on the chip, in the step's own order
CNN_AMD_SERIAL_BWD_GFLOP=0 python bench.py --config resnet18 --breakdown > $OUT/bench_resnet18_serial.json 2> $OUT/bench_resnet18_serial_breakdown.txt
# conv_rows / wgrad_sp on the stacks' shapes, isolated (tools/one_layer.py)
(for c in "256 64 112 112 128 3 1 0" "128 64 112 112 128 3 1 1" "128 128 56 56 256 3 1 1" "128 256 56 56 256 3 1 1" "128 256 28 28 512 3 1 1" "128 512 28 28 512 3 1 1" "128 512 14 14 512 3 1 1" "64 64 56 56 64 3 1 1" "64 128 28 28 128 3 1 1" "64 256 14 14 256 3 1 1" "64 512 7 7 512 3 1 1"; do echo "== $c"; python tools/one_layer.py $c 3 2>&1 | grep -v "amdgpu\|prep"; done) > $OUT/rows_sp_isolated.txt 2>&1
bash tools/run_tune.sh > /dev/null 2>&1; cp gpurun_out/tune_layers.log $OUT/layers_isolated.txt
(python tools/tune_stack.py vgg11; python tools/tune_stack.py resnet18) > $OUT/stack_layers_isolated.txt 2>&1
bash tools/pmc_conv_ns.sh $TAG > $OUT/hbm_traffic_conv_ns.txt 2>&1
# (round 6) the 3x3 / stride-2 layers isolated (conv_rows_s2 / conv_wgrad_sp2 against the kernels they replace), and where the first block's kernels wait
bash tools/s2_layers.sh > $OUT/s2_layers.txt 2>&1
bash tools/s2_wgrad_layers.sh > $OUT/s2_wgrad_layers.txt 2>&1
bash tools/any_layers.sh > $OUT/any_layers.txt 2>&1
bash tools/pmc_first_block.sh > $OUT/pmc_first_block.txt 2>&1; cp gpurun_out/pmc_first_block/summary.json $OUT/pmc_first_block.json 2>/dev/null
bash tools/timeline_cfg.sh rn resnet18 conv_stem_fwd > /dev/null 2>&1; cp gpurun_out/rn_timeline.txt $OUT/step_timeline_resnet18.txt 2>/dev/null
# large raw traces stay out of the merge-back (64 MiB cap): keep the per-kernel stats and drop the per-launch traces
find $OUT -name "*kernel_trace.csv" -size +8M -delete
find $OUT -name "*counter_collection.csv" -size +8M -delete
ls -R $OUT | head -60
