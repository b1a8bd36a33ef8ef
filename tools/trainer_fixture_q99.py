#!/usr/bin/env python3
"""dev tool (GPU box): the reference-run trainer fixture (tests/golden/train3_400_32x24.npz) through this path -- per step and
tensor, the 0.99 quantile and the maximum of |parameter - reference's parameter| in units of the tensor's learning rate,
over all elements and over the elements whose reference gradient is well above Adam's eps.  The numbers DESIGN.md section 3
and tests/test_gpu_optim.py::test_trainer_three_steps_match_reference quote.  usage: python tools/trainer_fixture_q99.py [out.json]"""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussian_splatting_amd import GaussianSplat
from gaussian_splatting_amd.trainer import make_optimizer, train_step, reference_regularised_mse, DEFAULT_LRS
fx = dict(np.load(os.path.join(ROOT, "tests", "golden", "train3_400_32x24.npz")))
dev = torch.device("cuda:0")
names = ["positions", "scales", "rotations", "opacity", "features"]
out = {}
for fused in (True, False):
    m = GaussianSplat(400)
    m.load_state_dict({k: torch.from_numpy(fx[k]) for k in names})
    m = m.to(dev)
    opt = make_optimizer(m, fused_clip_norm=0.1 if fused else None, write_clipped_grad=True)
    poses = torch.from_numpy(fx["poses"]).to(dev); target = torch.from_numpy(fx["target"]).to(dev)
    rows = {}
    for s in range(int(fx["steps"])):
        loss = train_step(m, opt, poses, target, reference_regularised_mse, max_grad_norm=0.1)
        for k in names:
            d = np.abs(getattr(m, k).detach().cpu().numpy() - fx[f"after{s + 1}_{k}"]).ravel() / DEFAULT_LRS[k]
            wc = np.abs(fx["clipped_g_" + k]).ravel() > 1e-6
            rows[f"step{s + 1}_{k}"] = {"q99_of_lr": round(float(np.quantile(d, 0.99)), 5), "max_of_lr": round(float(d.max()), 5),
                                         "well_conditioned_share": round(float(wc.mean()), 4),
                                         "well_conditioned_max_of_lr": round(float(d[wc].max()), 5) if wc.any() else 0.0}
        rows[f"step{s + 1}_loss_rel"] = abs(float(loss) - float(fx["losses"][s])) / abs(float(fx["losses"][s]))
    out["fused_clip_adam" if fused else "torch_adam"] = rows
worst = max(v["q99_of_lr"] for r in out.values() for k, v in r.items() if isinstance(v, dict))
worst_wc = max(v["well_conditioned_max_of_lr"] for r in out.values() for k, v in r.items() if isinstance(v, dict))
out["_summary"] = {"worst_q99_of_lr": worst, "worst_well_conditioned_max_of_lr": worst_wc,
                   "what": "|parameter - reference parameter| / lr after each of the reference's three trainer steps; cut 2^-20 (32x24 image)"}
print(json.dumps(out["_summary"]))
for k, v in out["fused_clip_adam"].items():
    print(k, v)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
