#!/usr/bin/env python3
"""Aggregates two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the bench into profiles/r1_pmc_hbm_traffic.json
(dev tool).  usage: tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import csv, json, sys, collections, re, os


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        acc[name].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out_path = sys.argv[3]
old = json.load(open(out_path)) if os.path.exists(out_path) else {}
kernels = {}
for k in sorted(fetch, key=lambda k: -(2 * fetch[k][1] + write.get(k, (0, 0))[1])):
    if not k.startswith("gsx::"):
        continue
    n, f = fetch[k]; w = write.get(k, (0, 0.0))[1]
    kernels[k] = {"launches": n, "FETCH_SIZE_KB_avg": round(f, 1), "WRITE_SIZE_KB_avg": round(w, 1),
                  "hbm_read_MB_corrected_x2": round(2 * f * 1024 / 2**20, 2), "hbm_write_MB": round(w * 1024 / 2**20, 2),
                  "hbm_total_MB": round((2 * f + w) * 1024 / 2**20, 2)}
res = {"_method": old.get("_method", ""), "kernels": kernels}
for k in old:
    if k.startswith("kernels_"):
        res[k] = old[k]
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps({k: v["hbm_total_MB"] for k, v in list(kernels.items())[:12]}, indent=1))
