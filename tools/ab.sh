#!/bin/bash
# dev tool (GPU box): A/B of library variants built by tools/build_variant.sh on ONE box: bench step time and stage times per variant
# usage: tools/ab.sh <name>[:ENV=VALUE] [<name> ...]   (restores the tree's own library afterwards)
cd /root/repo
L=gaussian-splatting_amd/csrc/libgsplat_gfx950.so
cp $L /tmp/lib_own.so
for spec in "$@"; do
  v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
  cp gpurun_variants/$v.so $L
  env $envs timeout -k 10 200 python3 bench.py --steps 120 --warmup 10 --no-cpu-baseline --no-inference-extra > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err || { echo "$v FAILED"; tail -3 gpurun_out/ab_$v.err; cp /tmp/lib_own.so $L; exit 1; }
  python3 -c "
import json,sys; d=json.load(open('gpurun_out/ab_$v.json')); print('$v', d['ms_per_step'], d['value'], {k:v for k,v in d['stage_ms'].items()})"
done
cp /tmp/lib_own.so $L
