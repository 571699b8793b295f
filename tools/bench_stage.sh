#!/bin/bash
# usage: tools/bench_stage.sh [ENV=VAL ...] -- prints value and per-stage ms of one bench run
env QI_TUNE=1 "$@" python bench.py --cpu-seconds 0 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_roofline']['device_ms_per_step'], d['step_roofline']['stage_ms_per_step'])"
