#!/bin/bash
# Collect the judged measurement artifacts on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh TAG   -> gpurun_out/TAG_*  (kernel traces as CSV, PMC passes of the roofline loop, bench lines)
TAG=${1:-r01_x}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
# single-stream trace: with the two encoder streams kernels overlap and their individual durations no longer add up to the step
TCCT_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_step -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${TAG}_step.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_roof -o roof -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $OUT/${TAG}_roof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_fetch -o fetch -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_write -o write -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $OUT/${TAG}_write.log 2>&1
# MFMA utilisation of the same loops (secondary evidence): busy cycles of the matrix pipes vs the GPU-active cycles
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_mfma -o mfma -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $OUT/${TAG}_mfma.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_sq -o sq -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only > $OUT/${TAG}_sq.log 2>&1
# the loss-side kernels of BASELINE configs[2..3] (gumbel column softmax, norm_add, the FPL multi-select): durations + HBM traffic of the same loops
TCCT_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stepfl -o stepfl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --los=di+reg+fpl > $OUT/${TAG}_stepfl.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_lroof -o lroof -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only --los=di+reg+fpl > $OUT/${TAG}_lroof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_lfetch -o lfetch -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only --los=di+reg+fpl > $OUT/${TAG}_lfetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_lwrite -o lwrite -- python3 $GRAFT_REPO_ROOT/bench.py --roofline-only --los=di+reg+fpl > $OUT/${TAG}_lwrite.log 2>&1
cd $GRAFT_REPO_ROOT
# the PMC traffic of the roofline kernels first (profiles/TAG_pmc.json in THIS copy of the tree), so that the bench lines below carry `roofline.traffic`
python tools/pmc_json.py ${TAG} > /dev/null 2>> $OUT/${TAG}_bench.err
python bench.py --steps 50 --warmup 10 > $OUT/${TAG}_bench.json 2>> $OUT/${TAG}_bench.err
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --los=di+reg > $OUT/${TAG}_bench_reg.json 2>> $OUT/${TAG}_bench.err
python bench.py --steps 30 --warmup 10 --no-cpu-baseline --los=di+reg+fpl > $OUT/${TAG}_bench_fullloss.json 2>> $OUT/${TAG}_bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --dtype fp32 > $OUT/${TAG}_bench_fp32.json 2>> $OUT/${TAG}_bench.err
python tools/infer_bench.py > $OUT/${TAG}_infer.txt 2>> $OUT/${TAG}_bench.err
python tools/infer_bench.py --bs 1 >> $OUT/${TAG}_infer.txt 2>> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json
# HBM traffic of the whole step (every kernel, single stream)
bash tools/step_traffic.sh ${TAG} > $OUT/${TAG}_traffic.txt 2>&1
# STEADY-STATE per-launch trace (3 warm-up + 3 traced steps, cut at the optimizer kernel): family table, launch-duration histogram, short launches by symbol
bash tools/steady_trace.sh ${TAG} > $OUT/${TAG}_steady.out 2>&1
rm -rf $OUT/${TAG}_steady
# round 6: the step by level and branch (marker launches; kernel trace + the two PMC passes), for --los=di, the full loss and KiteSeg.predict; the multi-stream step's
# level timeline from HIP events (no profiler); the allocator's plateau
bash tools/attrib_trace.sh ${TAG} > $OUT/${TAG}_attrib.log 2>&1
bash tools/attrib_trace.sh ${TAG}_fl --los di+reg+fpl >> $OUT/${TAG}_attrib.log 2>&1
bash tools/attrib_trace.sh ${TAG}_inf --infer >> $OUT/${TAG}_attrib.log 2>&1
python tools/level_timeline.py > $OUT/${TAG}_level_timeline.txt 2>> $OUT/${TAG}_bench.err
python tools/level_timeline.py --los di+reg+fpl > $OUT/${TAG}_level_timeline_fullloss.txt 2>> $OUT/${TAG}_bench.err
MEMGROW_STEPS=241 python tools/memgrow.py > $OUT/${TAG}_memgrow.txt 2>> $OUT/${TAG}_bench.err
rm -rf $OUT/${TAG}_attrib_trace $OUT/${TAG}_attrib_fetch $OUT/${TAG}_attrib_write $OUT/${TAG}_fl_attrib_trace $OUT/${TAG}_fl_attrib_fetch $OUT/${TAG}_fl_attrib_write $OUT/${TAG}_inf_attrib_trace $OUT/${TAG}_inf_attrib_fetch $OUT/${TAG}_inf_attrib_write
# keep only the small summaries
find $OUT/${TAG}_step $OUT/${TAG}_stepfl $OUT/${TAG}_roof $OUT/${TAG}_lroof $OUT/${TAG}_fetch $OUT/${TAG}_write $OUT/${TAG}_lfetch $OUT/${TAG}_lwrite $OUT/${TAG}_mfma $OUT/${TAG}_sq -type f ! -name '*kernel_stats.csv' ! -name '*counter_collection.csv' -delete 2>/dev/null
