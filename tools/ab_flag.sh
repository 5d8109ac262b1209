#!/bin/bash
# Same-box A/B of one bench.py flag: tools/ab_flag.sh "--wgrad-mode 4" [pairs] [extra bench.py args ...]
# runs `python bench.py FLAG` and `python bench.py` alternately (pairs times, default 3) and prints value / ms_per_step of each run.
flag="$1"; pairs="${2:-3}"; shift; shift
for i in $(seq 1 "$pairs"); do
  python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 10 $flag "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$flag', d['value'], d['ms_per_step'])" || exit 1
  python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'])" || exit 1
done
