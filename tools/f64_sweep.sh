# float64 engine timings (QI_TUNE knobs as arguments): tools/f64_sweep.sh
run() { echo -n "[$*] "; env QI_TUNE=1 "$@" python bench.py --dtype f64 --order 12 --channels 4 --cpu-seconds 0 --steps 3 --warmup 1 --wrappers 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_roofline']['stage_ms_per_step'])"; }
run3() { echo -n "[o3 $*] "; env QI_TUNE=1 "$@" python bench.py --dtype f64 --cpu-seconds 0 --steps 10 --warmup 2 --wrappers 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_roofline']['stage_ms_per_step'])"; }
run A=1
run3 A=1
run QI_TFR_LIB=$PWD/quantum_inferno_amd/libqi_tfr_u1w4.so
