#!/bin/bash
# dev tool (GPU box): runs a command with a library variant (tools/build_variant.sh) in place of the tree's own, then restores it.
# usage: tools/with_variant.sh <name> <command ...>
cd /root/repo
L=gaussian-splatting_amd/csrc/libgsplat_gfx950.so
v=$1; shift
cp $L /tmp/lib_own_wv.so; cp gpurun_variants/$v.so $L
"$@"; rc=$?
cp /tmp/lib_own_wv.so $L
exit $rc
