#!/bin/bash
# Collect SQ counters for the bench's kernels (separate passes, kernel-trace only; see MI355X guide).
# usage: tools/pmc.sh <outdir-under-gpurun_out> [bench args...]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/a -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --settle-ms 0 --cpu-seconds 0 "$@" > $out/a.log 2>&1
echo "pass a rc=$?"
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $out/b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --settle-ms 0 --cpu-seconds 0 "$@" > $out/b.log 2>&1
echo "pass b rc=$?"
