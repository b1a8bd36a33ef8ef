#!/bin/bash
# dev tool (GPU box): kernel trace of tools/soak.py; per-kernel average over the LAST 300 trainer steps' calls.  usage: tools/soak_ktrace.sh <tag> [steps]
tag=$1; steps=${2:-1200}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d /root/repo/gpurun_out/sk_$tag -o $tag --output-format csv -- python3 /root/repo/tools/soak.py $steps > /root/repo/gpurun_out/sk_$tag.log 2>&1 || { tail -5 /root/repo/gpurun_out/sk_$tag.log; exit 1; }
python3 - "$tag" <<'PY'
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f'/root/repo/gpurun_out/sk_{tag}/**/*kernel_trace.csv', recursive=True)
calls = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    calls[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
rows = []
for k, v in calls.items():
    if len(v) < 1200: continue
    per = len(v) // 1200 if len(v) >= 1200 else 1
    # forward-only renders and stage-timer passes come after the trainer steps: take the window before them
    first = v[: 200 * per]; last = v[900 * per: 1150 * per]
    rows.append((sum(last) / len(last) / 1e3, sum(first) / len(first) / 1e3, k))
rows.sort(reverse=True)
for l, fi, k in rows[:16]:
    print(f"{k[:70]:70s} first {fi:8.2f} us   late {l:8.2f} us")
PY
grep "ms/step" /root/repo/gpurun_out/sk_$tag.log
