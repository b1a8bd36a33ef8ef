#!/usr/bin/env python3
"""Distribution of active 8x8 quadrants per (tile, splat) instance on the bench scene (dev tool).
Vectorised restatement of quad_mask() (gsx_composite.hip) in torch; answers 'how much of the tile lists is
outside the ellipse' before changing the binning."""
import sys, os, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussian_splatting_amd import render as R
from gaussian_splatting_amd.scenes import config_scene, create_orbit_poses

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
views = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 13, 27, 44]
dev = torch.device("cuda:0")
params, H, W = config_scene(cfg)
P = tuple(torch.from_numpy(params[k]).to(dev) for k in ("positions", "scales", "rotations", "opacity", "features"))
poses = torch.from_numpy(create_orbit_poses(4.0, 60, 45.0)).to(dev)
tw = (W + 15) // 16


def lin(i, steps):
    step = 2.0 / (steps - 1)
    i = i.clamp(max=steps - 1).float()
    return torch.where(i < steps // 2, -1.0 + step * i, 1.0 - step * (steps - 1 - i))


def edge_max(kaa, kab, kbb, fixed, lo, hi):
    t = (-0.5 * kab * fixed / kaa)
    t = torch.minimum(torch.maximum(t, lo), hi)
    return kaa * t * t + kab * fixed * t + kbb * fixed * fixed


for v in views:
    outs, vs, extra = R.forward_view(P, poses[v], H, W)
    D = vs.D
    tiles = extra["tiles"][:D].long(); ranks = vs.ranks[:D].long()
    rec = vs.rec[ranks]
    mu_x, mu_y, k00, kb, k11, o = (rec[:, i] for i in range(6))
    bx = rec[:, 10].view(torch.int32).long(); by = rec[:, 11].view(torch.int32).long()
    qx = (tiles % tw) * 16; qy = (tiles // tw) * 16
    xlo, xhi = (bx & 0xffff) - qx, ((bx >> 16) & 0xffff) - qx
    ylo, yhi = (by & 0xffff) - qy, ((by >> 16) & 0xffff) - qy
    proper = (k00 < 0) & (k11 < 0) & (4 * k00 * k11 - kb * kb > 0)
    e = (((o.view(torch.int32) >> 23) & 0xff) - 127).clamp(min=-3)      # GSX_CUT_MIN_EXP
    thr = -(24 + 1 + e).float() - 0.0625
    cnt = torch.zeros(D, dtype=torch.long, device=dev)
    cnt_box = torch.zeros(D, dtype=torch.long, device=dev)
    for k in range(4):
        ox, oy = (k & 1) * 8, (k >> 1) * 8
        inbox = (xlo <= ox + 7) & (xhi >= ox) & (ylo <= oy + 7) & (yhi >= oy)
        ulo = lin(qx + ox, W) - mu_x; uhi = lin(qx + ox + 7, W) - mu_x
        vlo = lin(qy + oy, H) - mu_y; vhi = lin(qy + oy + 7, H) - mu_y
        inside = (ulo <= 0) & (uhi >= 0) & (vlo <= 0) & (vhi >= 0)
        qmax = torch.maximum(torch.maximum(edge_max(k00, kb, k11, vlo, ulo, uhi), edge_max(k00, kb, k11, vhi, ulo, uhi)),
                             torch.maximum(edge_max(k11, kb, k00, ulo, vlo, vhi), edge_max(k11, kb, k00, uhi, vlo, vhi)))
        qmax = torch.where(inside, torch.zeros_like(qmax), qmax)
        keep = inbox & ~(proper & (qmax < thr))
        cnt += keep.long(); cnt_box += inbox.long()
    hist = torch.bincount(cnt, minlength=5).tolist()
    hist_box = torch.bincount(cnt_box, minlength=5).tolist()
    print(json.dumps({"view": v, "D": D, "quadrants_active_hist": hist, "box_only_hist": hist_box,
                      "mean_active": round(float(cnt.float().mean()), 3)}))
    # lane utilisation: pixels of the splat's box inside the tile vs lanes spent (64 per active quadrant)
    ix = (torch.minimum(xhi, torch.full_like(xhi, 15)) - torch.maximum(xlo, torch.zeros_like(xlo)) + 1).clamp(min=0)
    iy = (torch.minimum(yhi, torch.full_like(yhi, 15)) - torch.maximum(ylo, torch.zeros_like(ylo)) + 1).clamp(min=0)
    area = (ix * iy).float()
    bw = ((bx >> 16) & 0xffff) - (bx & 0xffff) + 1; bh = ((by >> 16) & 0xffff) - (by & 0xffff) + 1
    r = vs.rec
    gbx = r[:, 10].view(torch.int32).long(); gby = r[:, 11].view(torch.int32).long()
    gw = (((gbx >> 16) & 0xffff) - (gbx & 0xffff) + 1).clamp(min=0).float(); gh = (((gby >> 16) & 0xffff) - (gby & 0xffff) + 1).clamp(min=0).float()
    vis = gw * gh > 0
    q = torch.tensor([0.1, 0.5, 0.9, 0.99], device=dev)
    print(json.dumps({"box_px_in_tile_mean": round(float(area.mean()), 1), "lanes_spent_mean": round(float(cnt.float().mean()) * 64, 1),
                      "box_fill": round(float(area.sum() / (cnt.float().sum() * 64)), 3),
                      "visible": int(vis.sum()), "box_w_quantiles": torch.quantile(gw[vis], q).tolist(), "box_h_quantiles": torch.quantile(gh[vis], q).tolist(),
                      "tiles_per_visible": round(D / max(int(vis.sum()), 1), 2)}))
