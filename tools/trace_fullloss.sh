#!/bin/bash
# Single-stream kernel trace of one --los=di+reg+fpl step (gpurun, repo root):  bash tools/trace_fullloss.sh TAG -> gpurun_out/TAG_trace/
TAG=${1:-tracefl}
LOS=${2:-di+reg+fpl}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
TCCT_STREAMS=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 2 --los=$LOS --no-cpu-baseline --no-roofline > $OUT/${TAG}_trace.log 2>&1
