#!/bin/bash
# dev tool (GPU box): per-kernel averages of tools/time_stages.py c3 for each library variant (tools/build_variant.sh); timing only --
# a variant may compute garbage.  usage: tools/ab_kstats.sh <pattern> <name> [<name> ...]
cd /root/repo
pat=$1; shift
L=gaussian-splatting_amd/csrc/libgsplat_gfx950.so
cp $L /tmp/lib_own.so
for v in "$@"; do
  cp gpurun_variants/$v.so $L
  echo "== $v"; tools/kstats.sh ab_$v c3 | grep -E "$pat" || true
done
cp /tmp/lib_own.so $L
