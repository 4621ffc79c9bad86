#!/bin/bash
# dev tool (CPU box): a library variant with extra compiler defines for EVERY kernel file, into
# hcatgnet_amd/csrc/_variants/<name>.so (git-ignored, travels to the GPU box; load with HCG_LIB, tools/ab_lib.sh / ab_kernels.sh).
# usage: tools/build_variant_def.sh NAME "-DHCG_SPLIT_SCALAR"        (an empty define string = a copy of the product build)
set -e
name=$1; defs=$2
C=/root/repo/hcatgnet_amd/csrc
mkdir -p $C/_variants /tmp/var_$name
files="api plan gemm layer pool fused mid wave tall readout head loss reduce optim collate"
n=0
for f in $files; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize $defs -c $C/$f.hip -o /tmp/var_$name/$f.o &
  n=$((n+1)); if [ $((n % 5)) -eq 0 ]; then wait; fi
done
wait
objs=""; for f in $files; do objs="$objs /tmp/var_$name/$f.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/_variants/$name.so $objs
echo built $C/_variants/$name.so
