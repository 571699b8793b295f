#!/bin/bash
# kernel statistics of the float64 leg (usage: tools/r4_prof_f64.sh <tag> [env assignments...])
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --legs f64 --cpu-seconds 0 --steps 20 --warmup 5 > $out/stats.log 2>&1
echo "stats rc=$?"
cd $GRAFT_REPO_ROOT
cp $(ls $out/stats/*/*kernel_stats.csv | tail -1) $out/kernel_stats.csv; rm -rf $out/stats
tail -1 $out/stats.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_roofline']['stage_ms_per_step'])"
head -24 $out/kernel_stats.csv | cut -d, -f1-4 | sed 's/qi::native::(anonymous namespace):://; s/(qi::native::[A-Za-z0-9]*)//' | cut -c1-150
