#!/bin/bash
# One round's evidence in one job (GPU box, from the repo root): rocprofv3 kernel statistics of the four bench legs, and per
# leg the TCC (FETCH_SIZE, WRITE_SIZE) and SQ_INSTS_VALU counter passes (separate --pmc runs with --kernel-trace only; the
# program goes directly after `--`), the SQ counter set of the float64 kernels and the per-item counts of the streaming leg.
# Results under gpurun_out/prof_<tag>/; tools/make_traffic_json.py turns them into profiles/traffic.json.
# usage: bash tools/profile_round.sh r05 [stats|pmc|all]
tag=${1:-r05}; what=${2:-all}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
declare -A LEG=( [cfg1]="--legs main --two-streams 0 --wrappers 0" [cfg2]="--legs configs2" [f64]="--legs f64" )
if [ "$what" != "pmc" ]; then
  bash tools/prof_leg.sh prof_$tag/cfg1 --legs main --steps 200 --warmup 20 --two-streams 0 --wrappers 0 > $out/cfg1.txt 2>&1; tail -9 $out/cfg1.txt
  bash tools/prof_leg.sh prof_$tag/cfg2 --legs configs2 --steps 10 --warmup 3 > $out/cfg2.txt 2>&1; tail -9 $out/cfg2.txt
  bash tools/prof_leg.sh prof_$tag/f64 --legs f64 --steps 20 --warmup 5 > $out/f64.txt 2>&1; tail -9 $out/f64.txt
  bash tools/prof_leg.sh prof_$tag/cfg4 --config 4 > $out/cfg4.txt 2>&1; tail -9 $out/cfg4.txt
fi
if [ "$what" != "stats" ]; then
  for leg in cfg1 cfg2 f64; do
    d=$out/pmc_$leg; mkdir -p $d
    for ctr in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
      ( cd /tmp && TMPDIR=/tmp timeout -k 10 500 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $d/$ctr -- python3 $GRAFT_REPO_ROOT/bench.py ${LEG[$leg]} --steps 3 --warmup 1 --settle-ms 0 --cpu-seconds 0 > $d/$ctr.log 2>&1 )
      echo "$leg $ctr rc=$?"
      rm -f $d/$ctr/*/*.db
    done
    python3 tools/traffic_summary.py $d > $out/${leg}_traffic_summary.txt 2>&1
  done
  bash tools/pmc_f64.sh prof_$tag/pmc_f64_sq > $out/pmc_f64_sq.log 2>&1; rm -rf $out/pmc_f64_sq/a $out/pmc_f64_sq/b
  bash tools/cfg4_pmc.sh prof_$tag/cfg4_pmc > $out/cfg4_pmc.log 2>&1; tail -3 $out/cfg4_pmc.log
  python3 tools/make_traffic_json.py $out $tag > $out/traffic.json; cat $out/traffic.json
fi
du -sh $out
