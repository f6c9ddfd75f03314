#!/bin/bash
# gemm256s.hip / attention64b.hip write M0 in their LDS-DMA asm without saving it: verify that hipcc emits no other M0 reference in
# those objects (an M0 use of the compiler's own between an M0 write and its DMA would break the destination address)
set -e
cd "$(dirname "$0")/../flux-2-swift-mlx_amd/csrc"
mkdir -p ../../gpurun_out/m0
for f in gemm256s attention64b attention64b_doff; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -S --offload-device-only $f.hip -o ../../gpurun_out/m0/$f.s
  n=$(grep -w "m0" ../../gpurun_out/m0/$f.s | grep -v "s_add_u32 m0, s\|s_mov_b32 m0, s" | grep -vc "^\s*;" || true)
  echo "$f: M0 references besides the LDS-DMA setup: $n"
  test "$n" = "0"
done
