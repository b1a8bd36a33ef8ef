"""GPU tier: every row of SURVEY 8 in one run, the way a user of the reference would chain them -- a teacher model renders
views (N1), they are written in the reference's on-disk scene layout and read back (N2), a perturbed student is trained on
them through the prefetcher and the one-call trainer step (a1-a13, N4), validation PSNR on held-out views rises, the
student's maintenance methods (N3) leave it renderable, and an orbit of it goes into a frame sink.  Nothing here is a
parity check (those are test_gpu_parity / test_gpu_optim); it guards the seams between the pieces."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _poses(n, seed):
    g = np.random.default_rng(seed)
    p = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    p[:, 0, 3] = g.uniform(-0.15, 0.15, n); p[:, 1, 3] = g.uniform(-0.15, 0.15, n); p[:, 2, 3] = g.uniform(1.8, 2.4, n)
    return p


def test_scene_on_disk_to_trained_student(tmp_path):
    from gaussian_splatting_amd import GaussianSplat
    from gaussian_splatting_amd import trainer as T
    from gaussian_splatting_amd.dataset import SceneDataset, DevicePrefetcher, write_scene
    from gaussian_splatting_amd.frames import FrameSink
    from gaussian_splatting_amd.scenes import synth_scene
    dev = torch.device("cuda:0")
    N, H, W = 3000, 96, 128
    sc = synth_scene(N, 21, 0.02, 0.06, o_lo=0.3, o_hi=0.9)
    teacher = GaussianSplat(N).to(dev)
    with torch.no_grad():
        for k, v in sc.items():
            getattr(teacher, k).copy_(torch.from_numpy(v))
    tr_poses, va_poses = _poses(12, 1), _poses(3, 2)
    frames_tr = teacher.render_frames(torch.from_numpy(tr_poses).to(dev), (H, W))
    frames_va = teacher.render_frames(torch.from_numpy(va_poses).to(dev), (H, W))
    assert frames_tr.shape == (12, 3, H, W) and float(frames_tr.mean()) > 0.05          # the scene is in view
    info = write_scene(tmp_path / "scene", {"train": (frames_tr, tr_poses), "val": (frames_va, va_poses)}, metadata={"scene_name": "teacher"})
    assert info["processing"]["splits"] == {"train": 12, "val": 3, "test": 0} and info["processing"]["image_size"] == [W, H]

    train, val = SceneDataset(tmp_path / "scene", "train"), SceneDataset(tmp_path / "scene", "val")
    assert len(train) == 12 and len(val) == 3
    first = next(iter(DevicePrefetcher(train, dev)))
    assert first[0].shape == (1, 3, H, W) and first[0].device == dev and first[1].shape == (1, 4, 4)
    assert (first[0][0] - frames_tr[0].clamp(0, 1)).abs().mean().item() < 0.02          # what was rendered, through JPEG and back

    student = GaussianSplat(N).to(dev)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for k, v in sc.items():
            getattr(student, k).copy_(torch.from_numpy(v))
        student.features.copy_(torch.rand(N, 3, generator=g))                           # forgets the colours
        student.opacity.mul_(0.7)
    opt = T.make_optimizer(student, lrs=dict(features=2e-2, opacity=5e-3), fused_clip_norm=0.1)
    before = T.evaluate(student, DevicePrefetcher(val, dev))
    losses = [T.train_epoch(student, opt, DevicePrefetcher(train, dev, order=np.random.default_rng(e).permutation(12)), T.l1_loss)
              for e in range(25)]
    assert getattr(opt, "_fused_step", None) is not None and opt._steps[student.positions] == 25 * 12      # ran as the one-call step
    after = T.evaluate(student, DevicePrefetcher(val, dev))
    assert before["views"] == after["views"] == 3
    assert losses[-1] < 0.6 * losses[0], losses
    assert after["psnr"] > before["psnr"] + 3.0 and after["l1"] < before["l1"], (before, after)

    n0 = student.num_gaussians
    student.prune_gaussians(0.05); student.optimize_scales(); student.merge_nearby_gaussians(0.002)
    assert 0 < student.num_gaussians <= n0
    pruned = T.evaluate(student, DevicePrefetcher(val, dev))
    assert np.isfinite(pruned["psnr"]) and pruned["psnr"] > before["psnr"]              # still the trained scene
    with FrameSink(tmp_path / "val_frames") as sink:
        assert student.render_frames(torch.from_numpy(va_poses).to(dev), (H, W), sink=sink) == 3
    assert sorted(p.name for p in (tmp_path / "val_frames").iterdir()) == [f"frame_{k:04d}.png" for k in range(3)]
