#!/bin/bash
# Multi-stream kernel trace of the default (3-stream) step: bash tools/trace_multi.sh TAG  -> gpurun_out/TAG_mtrace/ ; summarise with tools/busy_union.py
TAG=${1:-mt}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_mtrace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > $OUT/${TAG}_mtrace.log 2>&1
