#!/bin/bash
# dev tool: like tools/build_variant.sh, but every source is recompiled under the extra -D flags.  usage: tools/build_variant_all.sh <name> [-DX=Y ...]
set -e
name=$1; shift
cd "$(dirname "$0")/../gaussian-splatting_amd/csrc"
mkdir -p ../../gpurun_variants /tmp/variant_$name
F="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function"
for f in gsx_sort gsx_preprocess gsx_bin gsx_optim gsx_loss gsx_maint gsx_merge gsx_view gsx_comm; do
  /opt/rocm/bin/hipcc $F "$@" -c $f.hip -o /tmp/variant_$name/$f.o &
done
/opt/rocm/bin/hipcc $F -ffp-contract=fast -fno-slp-vectorize "$@" -c gsx_composite.hip -o /tmp/variant_$name/gsx_composite.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_variants/$name.so /tmp/variant_$name/*.o -ldl
echo built gpurun_variants/$name.so
