#!/bin/bash
# dev tool (GPU box): HBM traffic per launch (FETCH_SIZE x2, WRITE_SIZE: one --pmc pass each) of the bench's kernels for one library
# variant.  usage: tools/ab_traffic.sh <name> [ENV=VALUE ...]
cd /root/repo
v=$1; shift
for e in "$@"; do export "$e"; done
L=/root/repo/gaussian-splatting_amd/csrc/libgsplat_gfx950.so
cp $L /tmp/lib_own.so; cp gpurun_variants/$v.so $L
R=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 /root/repo/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-inference-extra"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $R/abt_${v}_fetch -o f --output-format csv -- $B > $R/abt_${v}_fetch.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $R/abt_${v}_write -o w --output-format csv -- $B > $R/abt_${v}_write.log 2>&1
rc=$?
cp /tmp/lib_own.so $L
python3 - $v <<'PY'
import csv, glob, sys, re, collections
v = sys.argv[1]
def load(pat, name):
    acc = collections.defaultdict(list)
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name: acc[re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")].append(float(r["Counter_Value"]))
    return acc
fe = load(f"/root/repo/gpurun_out/abt_{v}_fetch/**/*counter_collection.csv", "FETCH_SIZE")
wr = load(f"/root/repo/gpurun_out/abt_{v}_write/**/*counter_collection.csv", "WRITE_SIZE")
for k in sorted(fe, key=lambda k: -sum(fe[k]) / len(fe[k]))[:8]:
    r = 2 * sum(fe[k]) / len(fe[k]) / 1024; w = sum(wr.get(k, [0])) / max(len(wr.get(k, [0])), 1) / 1024
    print(f"{v} {k[:50]:50s} read x2 {r:8.1f} MB  write {w:8.1f} MB  total {r + w:8.1f} MB")
PY
exit $rc
