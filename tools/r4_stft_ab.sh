#!/bin/bash
# fused STFT variants in one job (configs[2] shape): product kernel, LDS budgets (segments per workgroup), 512-thread build, split kernel
export QI_TUNE=1
cd $GRAFT_REPO_ROOT
run() { echo -n "[$1] "; shift; env "$@" python tools/stft_bench.py 64 12 20 2>&1 | tail -1; }
run "fused G=8 (product)" QI_STFT_SPLIT=0
run "fused G=4, 3 workgroups per CU" QI_STFT_SPLIT=0 QI_STFT_LDS_KB=50
run "fused G=2, 6 workgroups per CU" QI_STFT_SPLIT=0 QI_STFT_LDS_KB=30
run "fused 512 threads, 128 registers" QI_STFT_SPLIT=0 QI_TFR_LIB=$PWD/quantum_inferno_amd/libqi_tfr_t512w4.so
run "split by bin parity, 16 segments" QI_STFT_SPLIT=1
run "fused G=8 (product)" QI_STFT_SPLIT=0
