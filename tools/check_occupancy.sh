#!/bin/bash
# Compile-time check (no GPU): waves per SIMD the compiler grants the hot kernels.  K1 (every k_is_valid_v5 instantiation) must stay at 2 --
# round 5 lost a third of Lift's rate when new code pushed the mesh-gate instantiation to 255 VGPRs + AGPRs = 1 wave.
cd "$(dirname "$0")/../mopa_rl_amd/csrc"
make -s resource-usage 2>&1 | grep -A9 "Function Name: _Z13k_is_valid_v5\|Function Name: _ZN4k3w[12]13k_rrt\|Function Name: _Z12k_env_dyn_ct\|Function Name: _Z10k_env_dyn4" |
  grep "Function Name\|VGPRs:\|AGPRs\|Occupancy\|VGPRs Spill\|Scratch" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | cut -c1-90
