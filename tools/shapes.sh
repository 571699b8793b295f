r() { echo -n "[$*] "; python bench.py "$@" --cpu-seconds 0 --wrappers 0 --two-streams 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['step_roofline']['frac'])"; }
r --channels 2
r --channels 4
r --channels 16
r --channels 64
r --order 12 --channels 1
r --order 12 --channels 8
r --config 2 --channels 16
r --log2n 18
r --log2n 22
r --log2n 16 --order 12 --steps 200
