"""Which stage entry points can be captured into a HIP graph?  (one torch.cuda.graph per stage)"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import render as R
from gaussian_splatting_amd.scenes import synth_scene, pose_tz
dev = torch.device("cuda:0")
params = synth_scene(20000, 3, 0.004, 0.03)
P = tuple(torch.from_numpy(params[k]).to(dev) for k in ("positions", "scales", "rotations", "opacity", "features"))
n, H, W = 20000, 256, 256
pose = torch.from_numpy(pose_tz(2.0)).to(dev)
R.forward_view(P, pose, H, W); torch.cuda.synchronize()
mean2d, depth, cov2d, keys, pre = R.project(*P, pose)
total = torch.zeros(2, dtype=torch.int64, device=dev)
order = R.depth_order(keys, 32, total[1:])
rec, tt, rank_of = R.build_records(pre, order, H, W)
offsets, total, _ = R.bin_count(tt, total)
cap = R._scratch.cap(dev, n)
tiles, ranks, ranges, chunk_base = R.bin_emit_sort(rec, offsets, total, n, H, W, cap)
torch.cuda.synchronize()
class _NoEvent:
    def record(self, *a): pass
    def synchronize(self): torch.cuda.synchronize()
_pinned = torch.zeros(2, dtype=torch.int64).pin_memory()
if len(sys.argv) > 1 and sys.argv[1] == "noevent":
    R._scratch.count_slot = lambda dev: (_pinned, _NoEvent())
stages = {
    "project": lambda: R.project(*P, pose),
    "depth_order": lambda: R.depth_order(keys, 32, total[1:]),
    "build_records": lambda: R.build_records(pre, order, H, W),
    "bin_count": lambda: R.bin_count(tt, total),
    "bin_emit_sort": lambda: R.bin_emit_sort(rec, offsets, total, n, H, W, cap),
    "composite_forward": lambda: R.composite_forward(rec, ranks, ranges, chunk_base, n, H, W, cap, True, True),
}
for name, fn in stages.items():
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            keep = fn()
        g.replay(); torch.cuda.synchronize()
        print(name, "captured and replayed", flush=True)
    except Exception as e:
        print(name, "FAILED:", str(e).splitlines()[0], flush=True)
        break
