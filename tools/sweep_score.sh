#!/bin/bash
# tuning sweep of the scoring kernel's launch shape (runs on the GPU box)
for vec in 1 2 4; do for kb in ${KBS:-1024 2048 3072 1000000}; do
  FY_SCORE_VEC=$vec FY_SCORE_TILE_KB=$kb python3 bench.py --steps 2 --warmup 1 --no-cpu --no-itemsim "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('vec=$vec tile_kb=$kb', 'ms/step %.1f'%d['ms_per_step'], {k: round(v,1) for k,v in d['phase_ms_rank0'].items()}, 'launches', d['roofline']['launches_per_step'])"
done; done
