#!/bin/bash
# the fused STFT of variant builds side by side in one job: tools/stft_ab.sh <variant|-> ...   ("-" = the product build)
for v in "$@"; do
  if [ "$v" = "-" ]; then unset QI_TFR_LIB; else export QI_TFR_LIB=$PWD/quantum_inferno_amd/libqi_tfr_$v.so; fi
  echo -n "[$v] "; python tools/stft_bench.py 64 12 20 2>&1 | tail -1
done
