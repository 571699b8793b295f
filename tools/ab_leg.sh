#!/bin/bash
# A/B of development switches on one bench leg, settings interleaved twice in ONE job (boxes differ by a few percent):
#   tools/ab_leg.sh "<bench.py args>" "ENV=a ENV2=b" "ENV=c" ...     e.g.  tools/ab_leg.sh "--legs f64" "QI_NATIVE_Z64_SLOTS=0" "QI_NATIVE_Z64_SLOTS=1"
args=$1; shift
for rep in 1 2; do
  for cfg in "$@"; do
    echo -n "[$cfg] "
    env QI_TUNE=1 $cfg python bench.py $args --cpu-seconds 0 --two-streams 0 --wrappers 0 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_roofline'].get('stage_ms_per_step'))" || exit 1
  done
done
