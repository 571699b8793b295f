#!/bin/bash
# round 4, first GPU pass: the new float64 tests, the result-path probe, the default bench line, float64 kernel statistics
out=$GRAFT_REPO_ROOT/gpurun_out/r4a
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "timed_shape or staged_result" > $out/tests.log 2>&1
echo "tests rc=$?"; tail -5 $out/tests.log
timeout -k 10 200 python tools/d2h_probe.py > $out/d2h_probe.txt 2>&1
echo "probe rc=$?"; cat $out/d2h_probe.txt
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; tail -c 1500 $out/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_f64 -- python3 $GRAFT_REPO_ROOT/bench.py --legs f64 --cpu-seconds 0 --steps 20 --warmup 5 > $out/stats_f64.log 2>&1
echo "stats f64 rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_cfg4 -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --cpu-seconds 0 > $out/stats_cfg4.log 2>&1
echo "stats cfg4 rc=$?"
cd $GRAFT_REPO_ROOT
for t in f64 cfg4; do cp $(ls $out/stats_$t/*/*kernel_stats.csv | tail -1) $out/${t}_kernel_stats.csv; rm -rf $out/stats_$t; done
ls -la $out
