#!/bin/bash
# dev tool (GPU box): one --pmc pass over a short bench run, per-launch averages of the kernels matching <pattern>.
# usage: tools/pmc_quick.sh <tag> <pattern> COUNTER [COUNTER ...]      (<= 8 SQ counters per pass)
tag=$1; pat=$2; shift 2
R=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc "$@" -d $R/pq_$tag -o q --output-format csv -- python3 /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-inference-extra > $R/pq_$tag.log 2>&1 || { tail -5 $R/pq_$tag.log; exit 1; }
python3 - "$R/pq_$tag" "$pat" <<'PY'
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
    if re.search(sys.argv[2], name):
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
