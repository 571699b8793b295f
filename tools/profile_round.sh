#!/bin/bash
# The round's evidence in one go (run on the GPU box from the repo root): the bench line, the rocprofv3 kernel-trace
# statistics of the SAME command, and the HBM traffic passes (TCC counters, one per pass).
# usage: tools/profile_round.sh <tag> [bench args, e.g. --config 2]      (the stage -> kernel map is below)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
python bench.py "$@" > $out/bench.json 2> $out/bench.err || exit 1
tail -1 $out/bench.json | cut -c1-600
key=$(tail -1 $out/bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']
import math
print('%s:%s:n%d:o%g:c%d' % (d['roofline']['kernel'], d['dtype'], int(math.log2(c['n'])), float(c['workload'].split('order N=')[1].split(',')[0]), c['channels_per_gpu']))")
stage=${key%%:*}
case $stage in block) kern="k_block_dual<";; zoom) kern="k_zoom2<";; pass2) kern="k_pass2<";; *) kern="k_";; esac
cd /tmp && export TMPDIR=/tmp
# (the same command without the CPU baseline and without the two-stream leg: its co-running steps stretch the kernels)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 --two-streams 0 "$@" > $out/stats.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
cp $(ls $out/stats/*/*kernel_stats.csv | tail -1) $out/kernel_stats.csv && rm -rf $out/stats
tools/traffic.sh traffic_$tag "$@" || exit 1
python tools/traffic_summary.py gpurun_out/traffic_$tag > $out/traffic_summary.txt
python tools/traffic_summary.py gpurun_out/traffic_$tag --json "$key" "$kern" | tail -1 > $out/traffic_dominant.json
for c in FETCH_SIZE WRITE_SIZE; do cp $(ls gpurun_out/traffic_$tag/$c/*/*counter_collection.csv | tail -1) $out/${c}_counter_collection.csv; done
rm -rf gpurun_out/traffic_$tag
echo "dominant stage key: $key"; cat $out/traffic_dominant.json
ls -la $out
