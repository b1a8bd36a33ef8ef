"""Probe: how much GPU time does replaying a view's forward as ONE HIP graph save over launching its ~20 kernels one by one?
(Static buffers, no instance-count check: a measurement, not a product path.)   python tools/graph_forward_probe.py [c3|c2]"""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import render as R
from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
dev = torch.device("cuda:0")
params, H, W = config_scene(cfg)
P = tuple(torch.from_numpy(params[k]).to(dev) for k in ("positions", "scales", "rotations", "opacity", "features"))
n = P[0].shape[0]
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
for v in range(0, 60, 5):
    R.forward_view(P, poses[v], H, W)                     # settles capacity and the depth-sort width
torch.cuda.synchronize()
reps = 40
for v in (0, 10):
    pose = poses[v].clone()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for mode in ("direct", "direct"):
        a.record()
        for _ in range(reps):
            att = R._enqueue_view_fused(P, pose, n, H, W, None, True, True)
        b.record(); torch.cuda.synchronize()
    t_direct = a.elapsed_time(b) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        att = R._enqueue_view_fused(P, pose, n, H, W, None, True, True)
    h_direct = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    R.STAGED = True
    class _NoEvent:
        def record(self, *a): pass
    _pinned = torch.zeros(2, dtype=torch.int64).pin_memory()
    _orig = R._scratch.count_slot
    R._scratch.count_slot = lambda dev: (_pinned, _NoEvent())          # an event record would invalidate the capture
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            att = R._enqueue_any(P, pose, n, H, W, None, True, True)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        att = R._enqueue_any(P, pose, n, H, W, None, True, True)
    R.STAGED = False
    R._scratch.count_slot = _orig
    torch.cuda.synchronize()
    for _ in range(2):
        a.record()
        for _ in range(reps):
            g.replay()
        b.record(); torch.cuda.synchronize()
    t_graph = a.elapsed_time(b) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    h_graph = (time.perf_counter() - t0) / reps * 1e6
    torch.cuda.synchronize()
    print(f"{cfg} view {v}: forward GPU time direct {t_direct*1e3:.1f} us (host {h_direct:.0f} us), graph {t_graph*1e3:.1f} us (host {h_graph:.0f} us), "
          f"D={int(_pinned[0])}", flush=True)
    del g
