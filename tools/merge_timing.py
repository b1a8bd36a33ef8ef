"""Times merge_nearby_gaussians' pair search (gsx_merge_count + scan + gsx_merge_fill) and the whole method on cuda:0.
usage: python tools/merge_timing.py [N] [threshold]"""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussian_splatting_amd import GaussianSplat

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.005
torch.manual_seed(0)
m = GaussianSplat(n).to("cuda:0")
m._merge_pairs_hip(thr)                      # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
off, partners = m._merge_pairs_hip(thr)
torch.cuda.synchronize(); t1 = time.perf_counter()
m.merge_nearby_gaussians(thr)
torch.cuda.synchronize(); t2 = time.perf_counter()
print(json.dumps({"N": n, "threshold": thr, "pairs": int(partners.numel()), "pair_search_ms": (t1 - t0) * 1e3,
                  "whole_method_ms": (t2 - t1) * 1e3, "n_after": m.num_gaussians,
                  "note": "pair_search includes the D2H copy of offsets and partners; the reference's torch.cdist would need N*N*4 bytes = %.1f TB" % (n * n * 4 / 1e12)}))
