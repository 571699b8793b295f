#!/bin/bash
# Round-4 evidence in one job (GPU box, from the repo root): kernel statistics of the four legs, HBM traffic and SQ counters
# of the float64 leg, instruction / byte counts per streamed item of configs[4].  Results under gpurun_out/prof_r04/.
out=$GRAFT_REPO_ROOT/gpurun_out/prof_r04
mkdir -p $out
cd $GRAFT_REPO_ROOT
bash tools/r4_prof_leg.sh prof_r04/cfg1 --legs main --steps 200 --warmup 20 --two-streams 0 --wrappers 0 > $out/cfg1.txt 2>&1; tail -12 $out/cfg1.txt
bash tools/r4_prof_leg.sh prof_r04/cfg2 --legs configs2 --steps 10 --warmup 3 > $out/cfg2.txt 2>&1; tail -12 $out/cfg2.txt
bash tools/r4_prof_leg.sh prof_r04/f64 --legs f64 --steps 20 --warmup 5 > $out/f64.txt 2>&1; tail -12 $out/f64.txt
bash tools/r4_prof_leg.sh prof_r04/cfg4 --config 4 > $out/cfg4.txt 2>&1; tail -12 $out/cfg4.txt
bash tools/traffic.sh prof_r04/traffic_f64 --legs f64 > $out/traffic_f64.log 2>&1
python tools/traffic_summary.py gpurun_out/prof_r04/traffic_f64 > $out/f64_traffic_summary.txt; tail -30 $out/f64_traffic_summary.txt
for c in FETCH_SIZE WRITE_SIZE; do cp $(ls $out/traffic_f64/$c/*/*counter_collection.csv | tail -1) $out/f64_${c}_counter_collection.csv; done
rm -rf $out/traffic_f64
bash tools/r4_pmc_f64.sh prof_r04/pmc_f64 > $out/pmc_f64.log 2>&1; rm -rf $out/pmc_f64/a $out/pmc_f64/b
bash tools/r4_cfg4_pmc.sh prof_r04/cfg4_pmc > $out/cfg4_pmc.log 2>&1; tail -3 $out/cfg4_pmc.log
ls -la $out
