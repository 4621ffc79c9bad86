#!/bin/bash
# dev tool (CPU box): library variants for A/B measurements into hcatgnet_amd/csrc/_variants/<name>.so (git-ignored, they travel
# to the GPU box).  usage: tools/build_variants.sh NAME FENCE   where FENCE is the asm text of mfma_results_fence,
# e.g.  tools/build_variants.sh fence20 's_nop 15\n\ts_nop 3'
set -e
name=$1; fence=$2
C=/root/repo/hcatgnet_amd/csrc
mkdir -p $C/_variants /tmp/var_$name
def="-DHCG_FENCE_ASM=\"$fence\""
for f in fused mid wave; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize "$def" -c $C/$f.hip -o /tmp/var_$name/$f.o &
done
wait
objs=""
for f in api plan gemm layer pool readout head loss reduce optim collate; do objs="$objs $C/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/_variants/$name.so $objs /tmp/var_$name/fused.o /tmp/var_$name/mid.o /tmp/var_$name/wave.o
echo built $C/_variants/$name.so
