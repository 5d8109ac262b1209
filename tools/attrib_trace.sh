#!/bin/bash
# Per-level / per-branch attribution of the single-stream training step (gpurun, repo root):
#   bash tools/attrib_trace.sh TAG [--los di+reg+fpl | --infer]  -> gpurun_out/TAG_attrib_summary.md, TAG_attrib_calls.csv
# Three runs of tools/attrib_trace.py (kernel trace, --pmc FETCH_SIZE, --pmc WRITE_SIZE: counters in their own passes), each with a marker launch in front of
# every C-ABI call; tools/attrib_summary.py cuts the dispatch sequences at the markers.
TAG=${1:-x}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_attrib_trace -o at -- python3 $GRAFT_REPO_ROOT/tools/attrib_trace.py --log $OUT/${TAG}_attrib_calls.json "$@" > $OUT/${TAG}_attrib_trace.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_attrib_fetch -o af -- python3 $GRAFT_REPO_ROOT/tools/attrib_trace.py --log $OUT/${TAG}_attrib_calls_f.json "$@" > $OUT/${TAG}_attrib_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_attrib_write -o aw -- python3 $GRAFT_REPO_ROOT/tools/attrib_trace.py --log $OUT/${TAG}_attrib_calls_w.json "$@" > $OUT/${TAG}_attrib_write.log 2>&1
rc=$?
find $OUT/${TAG}_attrib_trace $OUT/${TAG}_attrib_fetch $OUT/${TAG}_attrib_write -type f ! -name '*kernel_trace.csv' ! -name '*counter_collection.csv' -delete 2>/dev/null
cd $GRAFT_REPO_ROOT
python tools/attrib_summary.py $TAG > $OUT/${TAG}_attrib.out 2>&1
exit $rc
