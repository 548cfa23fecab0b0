#!/bin/bash
# Dump the gfx950 ISA of the kernels and a per-kernel resource summary (build container).
set -e
cd /root/repo/ndivplanning_amd/csrc
hipcc -O3 --offload-arch=gfx950 -std=c++17 -S --cuda-device-only -Wno-unused-value -Wno-pass-failed \
  -Rpass-analysis=kernel-resource-usage ndp_kernels.hip -o /tmp/ndp.s 2>&1 \
  | grep -E "Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy" | paste - - - - - \
  | sed 's/ndp_kernels.hip:[0-9]*:[0-9]*: remark: //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/Function Name: //; s/  */ /g' | cut -c1-160
