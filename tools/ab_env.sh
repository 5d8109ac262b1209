#!/bin/bash
# Same-box A/B of one environment switch: tools/ab_env.sh VAR=VALUE [pairs] [extra bench.py args ...]
# runs `VAR=VALUE python bench.py` and `python bench.py` alternately (pairs times, default 3) and prints value / ms_per_step of each run.
sw="$1"; pairs="${2:-3}"; shift; shift
for i in $(seq 1 "$pairs"); do
  env "$sw" python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sw', d['value'], d['ms_per_step'])" || exit 1
  python bench.py --no-cpu-baseline --no-roofline --steps 60 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'])" || exit 1
done
