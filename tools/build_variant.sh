#!/bin/bash
# dev tool: builds gpurun_variants/<name>.so = the library with gsx_composite.hip compiled under extra -D flags
# (A/B runs on one GPU box: copy the variant over csrc/libgsplat_gfx950.so inside the gpurun command).  usage: tools/build_variant.sh <name> [-DX=Y ...]
set -e
name=$1; shift
cd "$(dirname "$0")/../gaussian-splatting_amd/csrc"
mkdir -p ../../gpurun_variants
make -s all
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize "$@" -I. -c ${SRC:-gsx_composite.hip} -o /tmp/variant_$name.o
objs=$(ls *.o | grep -v gsx_composite.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_variants/$name.so $objs /tmp/variant_$name.o -ldl
echo built gpurun_variants/$name.so
