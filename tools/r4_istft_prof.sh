#!/bin/bash
# kernel times of the inverse ShortTimeFFT-convention transform, fused kernel against the three-kernel path (rocprofv3 --stats)
out=$GRAFT_REPO_ROOT/gpurun_out/istft_prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp QI_TUNE=1
for f in 1 0; do
  export QI_STFT_FUSED=$f
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/f$f -- python3 $GRAFT_REPO_ROOT/tools/istft_bench.py 16 2048 1024 > $out/f$f.log 2>&1
  echo "fused=$f rc=$?"; tail -1 $out/f$f.log
  python3 - <<PY
import csv,glob
f=sorted(glob.glob('$out/f$f/*/*kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:6]:
    print('   %-90s calls %5s avg %9.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3))
PY
  rm -rf $out/f$f
done
