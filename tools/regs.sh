#!/bin/bash
# Print VGPR / scratch / spill per kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage)
cd "$(dirname "$0")/../quantum_inferno_amd/csrc"
/opt/rocm/bin/hipcc -c -fPIC -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -I ../../include -I . ${1:-qi_native.hip} -o /tmp/regs.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "Function Name|VGPRs:|ScratchSize|VGPRs Spill" | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - - \
 | sed 's/Function Name: //; s/_ZN2qi6native12_GLOBAL__N_1//; s/EEEvNS0[^ \t]*//; s/ScratchSize \[bytes\/lane\]/scratch/'
