#!/bin/bash
# tools/quick_bench.sh "<ENV1=..,ENV2=..>" ... : one short RM2 bench line per environment variant -> phase times
for v in "$@"; do
  envs=$(echo "$v" | tr ',' ' ')
  echo "== $v"
  env $envs python3 bench.py --steps 3 --warmup 1 --no-cpu --no-itemsim --no-factorization --no-regime $QB_ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.2f' % d['ms_per_step'], {k: round(v, 2) for k, v in d['phase_ms_rank0'].items()}, 'roof %.3f' % d['roofline']['frac'])"
done
