#!/bin/bash
# STEADY-STATE per-launch kernel trace of the single-stream training step (gpurun, repo root):
#   bash tools/steady_trace.sh TAG [extra bench.py flags]  -> gpurun_out/TAG_steady/*kernel_trace.csv, summarised by tools/steady_trace.py TAG
# 3 warm-up + 3 timed steps; the summary keeps only the launches between the last four optimizer kernels (= the last three steps), so the model
# upload copies and the first steps' per-convolution weight packs are not in the table.
TAG=${1:-x}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
TCCT_STREAMS=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_steady -o st -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $OUT/${TAG}_steady.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/steady_trace.py $TAG
