run() { echo -n "[$*] "; env QI_TUNE=1 "$@" python bench.py --config 2 --channels 32 --cpu-seconds 0 --steps 4 --warmup 2 --settle-ms 250 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['stft']['ms_per_step'], d['stft']['frac'])"; }
run QI_STFT_LDS_KB=80
run QI_STFT_LDS_KB=40
run QI_STFT_LDS_KB=20
run QI_STFT_LDS_KB=150
