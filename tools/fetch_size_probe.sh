#!/bin/bash
# dev tool (GPU box): FETCH_SIZE / WRITE_SIZE of tools/fetch_size_probe.bin's four access shapes, per launch (KB), next to the bytes touched
R=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $R/fsp_fetch -o f --output-format csv -- /root/repo/tools/fetch_size_probe.bin > $R/fsp.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $R/fsp_write -o w --output-format csv -- /root/repo/tools/fetch_size_probe.bin >> $R/fsp.log 2>&1 || { tail -5 $R/fsp.log; exit 1; }
grep "bytes touched" $R/fsp.log | head -1
python3 - <<'PY'
import collections, csv, glob, re
for tag, ctr in (("fsp_fetch", "FETCH_SIZE"), ("fsp_write", "WRITE_SIZE")):
    f = glob.glob(f"/root/repo/gpurun_out/{tag}/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr:
            acc[re.sub(r"\(.*", "", r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{ctr:10s} {k:40s} {sum(v) / len(v):12.0f} KB per launch ({len(v)} launches)")
PY
