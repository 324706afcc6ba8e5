#!/bin/bash
# A/B of schedule variants: 30 timed steps each, two rounds interleaved (box-to-box variance is +-2 %, so only
# differences inside one call mean anything).  Usage: bash tools/ab_schedule.sh "VAR=a" "VAR=b VAR2=c" ...
VARIANTS=("X=0" "$@")
for round in 1 2; do
for v in "${VARIANTS[@]}"; do
  env $v python bench.py --steps 30 --warmup 3 --no-cpu --no-n65536 --config3 0 --config2 0 --config5 0 > gpurun_out/v.json 2>/dev/null
  python -c "
import json; d=json.loads(open('gpurun_out/v.json').read().strip().splitlines()[-1]); print('round $round', '$v', round(d['ms_per_step'],2), round(d['phases_ms_per_step']['factor_ms'],2), round(d['roofline']['achieved'],2))"
done; done
