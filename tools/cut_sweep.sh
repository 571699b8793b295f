# record-count threshold of the batch launch geometry (12 bands per block workgroup, long blocks): tools/cut_sweep.sh
r() { echo -n "[$*] "; env QI_TUNE=1 $ENVV python bench.py "$@" --cpu-seconds 0 --wrappers 0 --two-streams 0 --settle-ms 300 --steps 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_roofline']['stage_ms_per_step'])"; }
for C in 4 6; do for F in 4 8; do ENVV="QI_NATIVE_BLK_BATCH_FROM=$F"; echo -n "from=$F "; r --order 12 --channels $C; done; done
for C in 4 6; do for F in 4 8; do ENVV="QI_NATIVE_BLK_BATCH_FROM=$F"; echo -n "from=$F "; r --order 6 --channels $C; done; done
