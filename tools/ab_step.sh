#!/bin/bash
# same-box A/B of the whole step: bash tools/ab_step.sh "ENV_A=.." "ENV_B=.." [extra bench args]
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"; shift 2
for rep in 1 2; do
  for mode in "$A" "$B"; do
    env $mode python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], d['config']['step_ms_gpu_min_med_max'], d['value'])"
  done
done
