#!/bin/bash
# single-stream kernel traces of the training step under two environments, same box:  bash tools/trace_ab.sh TAG "ENV_A=.." "ENV_B=.." [bench args]
TAG=$1; A="$2"; B="$3"; shift 3
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for mode in "$A" "$B"; do
  i=$((i+1))
  export $mode
  TCCT_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_t$i -o step -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $OUT/${TAG}_t$i.log 2>&1
  unset ${mode%%=*}
  find $OUT/${TAG}_t$i -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
done
