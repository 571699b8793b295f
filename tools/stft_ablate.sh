#!/bin/bash
# timing ablations of the fused STFT (a -DQI_STFT_DBG build): tools/stft_ablate.sh
export QI_TUNE=1 QI_TFR_LIB=$PWD/quantum_inferno_amd/libqi_tfr_stftdbg.so
for d in 0 1 2 4 8 3 6 5 7 9 15; do
  echo -n "dbg=$d: "; QI_STFT_DBG=$d python tools/stft_bench.py 64 12 20 | tail -1
done
