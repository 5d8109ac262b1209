#!/bin/bash
# Per-launch kernel trace of one single-stream training step (gpurun, repo root):  bash tools/trace_step.sh TAG  -> gpurun_out/TAG_trace/
TAG=${1:-trace}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
TCCT_STREAMS=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${TAG}_trace.log 2>&1
