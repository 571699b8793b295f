#!/bin/bash
# per-kernel average durations of the one-record step (BASELINE configs[1]): rocprofv3 --kernel-trace --stats of tools/pair_probe.py
# usage (GPU box): bash tools/step_trace.sh [tag] [channels]   -> gpurun_out/<tag>/{probe.txt,kernel_stats.csv,stats.txt}
tag=${1:-step}; ch=${2:-1}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
QI_TUNE=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/tools/pair_probe.py $ch 0:0:0 > $out/probe.txt 2>&1
rc=$?
cd $GRAFT_REPO_ROOT
cat $out/probe.txt | grep -v amdgpu.ids
f=$(ls $out/trace/*/*kernel_stats.csv 2>/dev/null | tail -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv && python3 - "$out/kernel_stats.csv" <<'PY' | tee $out/stats.txt
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_\w+<[^(]*>|k_\w+)", r["Name"])
    if m and int(r["Calls"]) > 100:
        print(f"{m.group(1)[:60]:62s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e3:8.2f} us  {float(r['Percentage']):5.1f} %")
PY
rm -rf $out/trace
exit $rc
