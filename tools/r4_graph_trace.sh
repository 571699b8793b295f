#!/bin/bash
# kernel timeline of qi_cwt_stx as a captured graph (one record): do the block branch and the zoom chain overlap?
out=$GRAFT_REPO_ROOT/gpurun_out/r4_graph
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/tools/graph_probe.py 1 > $out/probe.txt 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $out/trace/*/*kernel_trace.csv | tail -1)
python tools/timeline_csv.py $f > $out/timeline.txt 2>&1
tail -5 $out/probe.txt
python - "$f" <<'PY' | tee $out/overlap.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 40 kernels of the run = graph mode steps
last = rows[-42:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    name = r["Kernel_Name"].split("(")[0].split("::")[-1][:40]
    print(f"{name:42s} start {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  end {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}")
PY
rm -rf $out/trace
