#!/bin/bash
# HBM traffic of the bench's kernels from the TCC counters, one counter per pass (MI355X guide: FETCH_SIZE
# and WRITE_SIZE do not fit one pass; kernel-trace only).  usage: tools/traffic.sh <outdir-under-gpurun_out> [bench args]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $out
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/$ctr -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --settle-ms 0 --cpu-seconds 0 --two-streams 0 "$@" > $out/$ctr.log 2>&1
  echo "$ctr rc=$?"
done
