#!/bin/bash
# gemm256.hip writes M0 in its LDS-DMA asm without saving it: verify that hipcc emits no other M0 reference in that object
set -e
cd "$(dirname "$0")/../flux-2-swift-mlx_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -c gemm256.hip -o /tmp/gemm256_m0.o -save-temps=obj -w
n=$(grep -w "m0" /tmp/gemm256-hip-amdgcn-amd-amdhsa-gfx950.s | grep -v "s_mov_b32 m0, s" | grep -vc "^\s*;" || true)
echo "M0 references besides the LDS-DMA setup: $n"
test "$n" = "0"
