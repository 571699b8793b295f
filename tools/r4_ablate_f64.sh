#!/bin/bash
# timing ablations of the float64 fine kernel (debug library): per-kernel averages of k_z64_fine for each QI_NATIVE_DEBUG mask
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_ablate}
mkdir -p $out
export QI_TFR_LIB=$GRAFT_REPO_ROOT/quantum_inferno_amd/libqi_tfr_dbg.so QI_TUNE=1
cd /tmp && export TMPDIR=/tmp
for mask in 0 1 2 4 8 16 5 13 15 31; do
  export QI_NATIVE_DEBUG=$mask
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/s$mask -- python3 $GRAFT_REPO_ROOT/bench.py --legs f64 --cpu-seconds 0 --steps 10 --warmup 3 --settle-ms 500 > $out/s$mask.log 2>&1
  f=$(ls $out/s$mask/*/*kernel_stats.csv | tail -1)
  python3 - "$f" $mask <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
pick={}
for r in rows:
    n=r['Name']
    if 'k_z64_fine' in n:
        key=n.split('k_z64_fine')[1].split('(')[0]
        pick[key]=float(r['AverageNs'])/1e3
tot=sum(pick.values())
print('mask',sys.argv[2],'fine total us %.0f'%tot,' '.join(f"{k}:{v:.0f}" for k,v in sorted(pick.items())))
PY
  rm -rf $out/s$mask
done
