#!/bin/bash
# VGPR / LDS / scratch per kernel (hipcc -Rpass-analysis=kernel-resource-usage).
cd "$(dirname "$0")/../pearl_amd/csrc"
for f in arena.hip dqn.hip; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/_kr.o 2>&1 |
    grep -E "Function Name|    VGPRs:|AGPRs|ScratchSize|LDS Size|Occupancy" |
    sed -E 's/.*remark: [^ ]+ +//; s/ \[-Rpass.*//' | paste - - - - - - | sed -E 's/Function Name: //; s/\t/ | /g'
done
