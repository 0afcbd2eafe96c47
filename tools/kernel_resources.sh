#!/bin/bash
# usage: tools_ru.sh file.hip  -> per-kernel VGPR / scratch / occupancy
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage --offload-device-only 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy" | sed -E 's/.*remark: +//; s/ *\[-Rpass.*//' | paste - - - - - - | sed -E 's/Function Name: //'
