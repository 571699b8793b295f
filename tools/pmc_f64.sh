#!/bin/bash
# SQ counters of the float64 leg's kernels (two passes) + the plan's band assignment
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
A="python3 $GRAFT_REPO_ROOT/bench.py --legs f64 --cpu-seconds 0 --steps 3 --warmup 1 --settle-ms 0"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $out/a -- $A > $out/a.log 2>&1
echo "pass a rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $out/b -- $A > $out/b.log 2>&1
echo "pass b rc=$?"
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py k_z64,k_block64 $out/a $out/b > $out/sq_counters.txt
QI_TUNE=1 QI_NATIVE_VERBOSE=1 python -c "
import torch, numpy as np
import quantum_inferno_amd as qi
n, fs, order = 1 << 20, 1000.0, 12
p = qi.TfrPlan(n, torch.float64, 'cuda:0', qi.TfrPlan.workspace_for(n, 167, torch.float64, 1))
p.set_styx_bank(order, fs); p.set_stx_bands(order, fs)
" 2> $out/plan_verbose.txt
grep -i "float64 zoom\|class" $out/plan_verbose.txt | head
rm -rf $out/a/*/*.db $out/b/*/*.db
wc -l $out/sq_counters.txt
