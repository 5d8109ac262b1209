#!/bin/bash
# factorised-attention variant at the bench shape: bench line + single-stream kernel trace (run through gpurun from the repo root)
TAG=${1:-r01_fa}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python bench.py --att factor --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
TCCT_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_step -o step -- python3 $GRAFT_REPO_ROOT/bench.py --att factor --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${TAG}_step.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
find $OUT/${TAG}_step -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
cat $OUT/${TAG}_bench.json
