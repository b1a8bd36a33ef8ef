#!/usr/bin/env python3
"""Aggregates the SQ counter passes of tools/pmc_sq.sh into profiles/r1_pmc_sq_counters.json (dev tool): per kernel, per-launch
averages and the derived VALU issue share = SQ_INSTS_VALU x 4 cycles / (SQ_BUSY_CYCLES-based kernel cycles x SIMDs)."""
import csv, json, sys, collections, re

def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if name.startswith("gsx::"):
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc

a = load(sys.argv[1]); b = load(sys.argv[2])
out = {}
for k in a:
    row = {c: round(sum(v) / len(v), 1) for c, v in a[k].items()}
    row.update({c: round(sum(v) / len(v), 1) for c, v in b.get(k, {}).items()})
    row["launches"] = len(next(iter(a[k].values())))
    out[k] = row
order = sorted(out, key=lambda k: -out[k].get("SQ_INSTS_VALU", 0))
res = {"_method": "rocprofv3 --pmc, two passes (tools/pmc_sq.sh) over `python3 bench.py --steps 60 --warmup 5 --no-cpu-baseline`, round-2 build (cut 2^-20), c3 scene; "
                  "per-launch averages summed over the chip. SQ_INSTS_* count wave-level instructions; a wave64 VALU instruction occupies its SIMD's issue port for "
                  "4 cycles; 256 CUs x 4 SIMDs. valu_issue_share = SQ_INSTS_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024) (the counter is summed over the 8 XCDs: 5.42 M for the 281-us backward = 8 x 678 k cycles at 2.4 GHz): the share of all VALU issue slots of the chip "
                  "that the kernel filled while it ran (GRBM_GUI_ACTIVE = cycles the kernel kept the GPU busy, as rocprofv3 reports it per dispatch).",
       "kernels": {}}
for k in order:
    r = out[k]
    gui = r.get("GRBM_GUI_ACTIVE")
    if gui and r.get("SQ_INSTS_VALU"):
        r["valu_issue_share"] = round(r["SQ_INSTS_VALU"] * 4 / (gui / 8 * 1024), 3)      # GRBM_GUI_ACTIVE comes summed over the 8 XCDs
    if r.get("SQ_WAVES"):
        r["valu_insts_per_wave"] = round(r.get("SQ_INSTS_VALU", 0) / r["SQ_WAVES"], 1)
        r["lds_insts_per_wave"] = round(r.get("SQ_INSTS_LDS", 0) / r["SQ_WAVES"], 1)
    res["kernels"][k] = r
json.dump(res, open(sys.argv[3], "w"), indent=1)
for k in order[:8]:
    r = res["kernels"][k]
    print(k[:44], {x: r.get(x) for x in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVES", "GRBM_GUI_ACTIVE", "valu_issue_share", "SQ_LDS_BANK_CONFLICT")})
