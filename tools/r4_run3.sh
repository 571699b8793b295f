#!/bin/bash
# float64 parity tests, then kernel statistics of the float64 leg (usage: tools/r4_run3.sh <tag> [pytest -k expr])
tag=$1; expr=${2:-"float64"}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$expr" > $out/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -6 $out/tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/r4_prof_f64.sh $tag > $out/prof.txt 2>&1; head -3 $out/prof.txt
python - <<PY
import csv
rows=list(csv.DictReader(open('$out/kernel_stats.csv')))
for r in rows[:34]:
    name=r['Name'].replace('qi::native::(anonymous namespace)::','').replace('void ','').split('(')[0][:60]
    print(f"{name:62s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.1f}")
PY
