#!/bin/bash
# dev tool (GPU box): per-kernel averages of the BENCH's one-call trainer step (rocprofv3 kernel trace of bench.py) for each library
# variant.  usage: tools/ab_bench_kstats.sh <pattern> <name> [<name> ...]
cd /root/repo
pat=$1; shift
L=gaussian-splatting_amd/csrc/libgsplat_gfx950.so
cp $L /tmp/lib_own_bk.so
for v in "$@"; do
  cp gpurun_variants/$v.so $L
  echo "== $v"
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/bk_$v -o $v --output-format csv -- python3 /root/repo/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-inference-extra > /root/repo/gpurun_out/bk_$v.log 2>&1 )
  python3 - "$v" "$pat" <<'PY'
import csv, glob, sys, re
v, pat = sys.argv[1], sys.argv[2]
f = glob.glob(f'/root/repo/gpurun_out/bk_{v}/**/*kernel_stats.csv', recursive=True)
tot = 0.0
for r in csv.DictReader(open(f[0])):
    if 'gsx::' in r['Name']: tot += float(r['TotalDurationNs'])
    if re.search(pat, r['Name']): print(f"{r['Name'][:60]:60s} {r['Calls']:>5s} {float(r['AverageNs'])/1000:8.2f} us")
print(f"all gsx kernels: {tot/1e6:.2f} ms")
PY
done
cp /tmp/lib_own_bk.so $L
