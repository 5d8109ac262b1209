#!/bin/bash
# HBM traffic of the WHOLE training step from the PMC counters (one pass per counter, single stream so that per-kernel numbers are attributable):
#   bash tools/step_traffic.sh TAG  ->  gpurun_out/TAG_sfetch / TAG_swrite (counter_collection.csv); summarise with tools/step_traffic.py TAG
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
TCCT_STREAMS=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_sfetch -o sfetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${TAG}_sfetch.log 2>&1
TCCT_STREAMS=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_swrite -o swrite -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${TAG}_swrite.log 2>&1
find $OUT/${TAG}_sfetch $OUT/${TAG}_swrite -type f ! -name '*counter_collection.csv' -delete 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/step_traffic.py $TAG
