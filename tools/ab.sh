#!/bin/bash
# usage: tools/ab.sh "ENV=VAL ..." "ENV=VAL ..." ...  -- steady-state bench (500 steps after 100) of each setting, twice, interleaved
for rep in 1 2; do
  for cfg in "$@"; do
    echo -n "[$cfg] "
    env QI_TUNE=1 $cfg python bench.py --cpu-seconds 0 --steps 500 --warmup 100 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_roofline']['stage_ms_per_step'])" || exit 1
  done
done
