#!/bin/bash
# dev tool: kernel-trace the bench and print the per-kernel averages (run on the GPU box through gpurun)
# usage: tools/prof_kernels.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_$tag -o $tag --output-format csv -- python3 /root/repo/bench.py --no-cpu-baseline --steps 60 "$@" > /root/repo/gpurun_out/prof_$tag.log 2>&1
python3 - "$tag" <<'PY'
import csv, glob, sys, re
tag = sys.argv[1]
f = glob.glob(f'/root/repo/gpurun_out/prof_{tag}/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:30]:
    print(f"{r['Name'][:56]:56s} {r['Calls']:>5s} {float(r['AverageNs'])/1000:8.2f} us {float(r['Percentage']):6.2f}%")
log = open(f'/root/repo/gpurun_out/prof_{tag}.log').read()
m = re.search(r'"ms_per_step": ([0-9.]+)', log)
print("ms_per_step under rocprof:", m.group(1) if m else None)
PY
