#!/bin/sh
# Builds the probe libraries next to their sources (they travel to the GPU box with the snapshot; *.so is git-ignored).
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o libclock_probe.so clock_probe.hip
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 wave_sum_probe.hip -o wave_sum_probe    # ./wave_sum_probe on the GPU box: DPP / permlane butterfly vs __shfl_xor, bit for bit
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value issue_overlap_probe.hip -o issue_overlap_probe    # ./issue_overlap_probe: does a wave's vector-memory issue slow its SIMD partner's MFMAs?
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value mfma_power.hip -o mfma_power    # ./mfma_power: sustained TFLOP/s of the two fp16 MFMA shapes (and of 16x16x32 with LDS fragment reads) under the power cap
