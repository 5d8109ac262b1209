#!/bin/bash
# VALU / wait picture of every kernel of the training step (single stream): one PMC pass with SQ_WAVE_CYCLES, SQ_BUSY_CYCLES, SQ_ACTIVE_INST_VALU, SQ_INSTS_VALU
#   bash tools/step_valu.sh TAG  ->  gpurun_out/TAG_svalu/svalu_counter_collection.csv ; summarise with tools/step_valu.py TAG
TAG=${1:-x}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
TCCT_STREAMS=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_svalu -o svalu -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${TAG}_svalu.log 2>&1
find $OUT/${TAG}_svalu -type f ! -name '*counter_collection.csv' -delete 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/step_valu.py $TAG
