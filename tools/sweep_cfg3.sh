# Same-box A/B sweeps of QI_TUNE knobs (run on the GPU box: bash tools/sweep_cfg3.sh).  `run` = configs[2] at 32 records,
# `run1` = configs[1], `run16` = configs[1] x 16 records; put the knob settings to compare at the end of this file.
run() { echo -n "[$*] "; env QI_TUNE=1 "$@" python bench.py --config 2 --channels 32 --cpu-seconds 0 --steps 8 --warmup 2 --settle-ms 250 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['step_roofline']['frac'], d['step_roofline']['stage_ms_per_step'], d['stft']['ms_per_step'])"; }
run1() { echo -n "[cfg1 $*] "; env QI_TUNE=1 "$@" python bench.py --cpu-seconds 0 --steps 300 --warmup 50 --settle-ms 250 --wrappers 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_roofline']['stage_ms_per_step'])"; }
run16() { echo -n "[cfg1x16 $*] "; env QI_TUNE=1 "$@" python bench.py --channels 16 --cpu-seconds 0 --steps 50 --warmup 10 --settle-ms 250 --wrappers 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_roofline']['stage_ms_per_step'])"; }
run1 A=1
run A=1
