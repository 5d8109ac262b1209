#!/bin/bash
# same-box A/B/C of the whole step: bash tools/ab3.sh "ENV_A=.." "ENV_B=.." "ENV_C=.." (two interleaved repetitions)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for mode in "$@"; do
    env $mode python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], d['config']['step_ms_gpu_min_med_max'], d['value'])"
  done
done
