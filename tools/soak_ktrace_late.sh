#!/bin/bash
# dev tool (GPU box): kernel trace of tools/soak.py <steps>; per-kernel average over each kernel's calls in the LAST 300 trainer steps
# (the forward-only renders and stage-timer passes behind them are cut off by call count).  usage: tools/soak_ktrace_late.sh <tag> <steps>
tag=$1; steps=${2:-2100}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/skl_$tag -o $tag --output-format csv -- python3 /root/repo/tools/soak.py $steps > /root/repo/gpurun_out/skl_$tag.log 2>&1 || { tail -5 /root/repo/gpurun_out/skl_$tag.log; exit 1; }
python3 - "$tag" "$steps" <<'PY'
import csv, glob, sys, collections
tag, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(f'/root/repo/gpurun_out/skl_{tag}/**/*kernel_trace.csv', recursive=True)
calls = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    calls[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
rows = []
for k, v in calls.items():
    if len(v) < steps: continue
    per = len(v) // steps
    w = v[(steps - 300) * per: (steps - 20) * per]
    rows.append((sum(w) / len(w) / 1e3, len(v), k))
rows.sort(reverse=True)
for t, n, k in rows[:14]:
    print(f"{k[:80]:80s} calls {n:7d}  late avg {t:9.2f} us")
PY
grep "ms/step" /root/repo/gpurun_out/skl_$tag.log | cut -c1-70
