#!/bin/bash
# dev tool (GPU box): per-kernel averages (rocprofv3 kernel trace) of tools/time_stages.py on a config.  usage: tools/kstats.sh <tag> [cfg] [views]
tag=$1; cfg=${2:-c3}; views=${3:-}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/ks_$tag -o $tag --output-format csv -- python3 /root/repo/tools/time_stages.py $cfg $views > /root/repo/gpurun_out/ks_$tag.log 2>&1 || { tail -5 /root/repo/gpurun_out/ks_$tag.log; exit 1; }
python3 - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f'/root/repo/gpurun_out/ks_{tag}/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print(f"{r['Name'][:64]:64s} {r['Calls']:>5s} {float(r['AverageNs'])/1000:8.2f} us {float(r['Percentage']):6.2f}%")
PY
tail -1 /root/repo/gpurun_out/ks_$tag.log | cut -c1-400
