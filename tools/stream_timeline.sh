#!/bin/bash
# per-launch kernel trace of the training step WITH its side streams (gpurun, repo root): bash tools/stream_timeline.sh TAG -> gpurun_out/TAG_mstream_summary.md
TAG=${1:-x}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_mstream -o ms -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-roofline "$@" > $OUT/${TAG}_mstream.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/stream_timeline.py $TAG
