#!/bin/bash
# The round's evidence in one go (run on the GPU box from the repo root): the default bench line, the rocprofv3
# kernel-trace statistics of the SAME command, and the HBM traffic passes.  usage: tools/profile_round.sh <tag>
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err || exit 1
tail -1 $out/bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-seconds 0 > $out/stats.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
cp $(ls $out/stats/*/*kernel_stats.csv | tail -1) $out/kernel_stats.csv && rm -rf $out/stats
tools/traffic.sh traffic_$tag || exit 1
python tools/traffic_summary.py gpurun_out/traffic_$tag > $out/traffic_summary.txt
# (the bench step is qi_cwt_stx: its joint launches are k_zoom2 / k_block_dual)
python tools/traffic_summary.py gpurun_out/traffic_$tag --json "zoom:f32:n20:o3:c1" "k_zoom2<" | tail -1 > $out/traffic_zoom.json
python tools/traffic_summary.py gpurun_out/traffic_$tag --json "block:f32:n20:o3:c1" "k_block_dual<" | tail -1 > $out/traffic_block.json
for c in FETCH_SIZE WRITE_SIZE; do cp $(ls gpurun_out/traffic_$tag/$c/*/*counter_collection.csv | tail -1) $out/${c}_counter_collection.csv; done
rm -rf gpurun_out/traffic_$tag
ls -la $out
