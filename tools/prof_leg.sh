#!/bin/bash
# kernel statistics of one bench leg: tools/prof_leg.sh <tag> <bench args...>
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 "$@" > $out/stats.log 2>&1
echo "stats rc=$?"
cd $GRAFT_REPO_ROOT
cp $(ls $out/stats/*/*kernel_stats.csv | tail -1) $out/kernel_stats.csv; rm -rf $out/stats
grep '^{' $out/stats.log | tail -1 > $out/bench.json
python - <<PY
import json,csv
d=json.loads(open('$out/bench.json').read())
print('value', d['value'], 'ms/step', d['ms_per_step'], d.get('step_roofline',{}).get('stage_ms_per_step'))
rows=list(csv.DictReader(open('$out/kernel_stats.csv')))
steps=d['steps']+d['warmup']
agg={}
for r in rows:
    n=r['Name'].replace('qi::native::(anonymous namespace)::','').replace('void ','').split('(')[0]
    key='hipfft' if (n.startswith('fft_rtc') or n.startswith('transpose_rtc')) else n.split('<')[0]
    agg[key]=agg.get(key,0)+float(r['TotalDurationNs'])
tot=sum(agg.values())
for k,v in sorted(agg.items(), key=lambda kv:-kv[1])[:14]:
    print(f"  {k:28s} {v/1e6:9.1f} ms total  {100*v/tot:5.1f} %")
PY
