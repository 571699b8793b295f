#!/bin/bash
# round 4: float64 tests + float64 bench legs (after a kernel change)
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4b}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "float64 or f64 or stream or staged_result" > $out/tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -15 $out/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --legs f64,configs4 --steps 20 --warmup 5 --cpu-seconds 0 > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"; python tools/show_bench.py $out/bench.json 2>/dev/null | head -40 || tail -c 1500 $out/bench.json
