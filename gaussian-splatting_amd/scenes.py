"""Synthetic scenes and camera paths of the BASELINE configs, numpy PCG64 so that
they are identical on every machine, plus the orbit camera path of the reference's visualiser
(/root/reference/utils/visualization.py:132-178)."""
import numpy as np


def synth_scene(n, seed, s_lo, s_hi, zq=None, o_lo=0.0, o_hi=1.0):
    """p.xy~U(-1,1), p.z~U(0.5,1.5) (keeps the world-z Jacobian bounded), s log-uniform in
    [s_lo, s_hi], q~N(0,1)^4 normalised, o~U(o_lo,o_hi), c~U(0,1).  zq quantises z so depths tie."""
    g = np.random.default_rng(seed)
    p = np.empty((n, 3), np.float32)
    p[:, :2] = g.random((n, 2), np.float32) * 2 - 1
    p[:, 2] = g.random(n, np.float32) + 0.5
    if zq:
        p[:, 2] = np.round(p[:, 2] * zq) / zq
    s = np.exp(g.random((n, 3), np.float32) * (np.log(s_hi) - np.log(s_lo)) + np.log(s_lo)).astype(np.float32)
    q = g.standard_normal((n, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    o = (g.random((n, 1), np.float32) * (o_hi - o_lo) + o_lo).astype(np.float32)
    c = g.random((n, 3), np.float32)
    return dict(positions=p, scales=s, rotations=q.astype(np.float32), opacity=o, features=c)


CONFIGS = {
    # name: (N, H, W, seed, s_lo, s_hi, zq)
    "c2": (100_000, 800, 800, 2, 0.002, 0.01, None),
    "c3": (1_000_000, 1024, 1024, 5, 0.001, 0.005, None),
    "c4": (3_000_000, 1024, 1024, 6, 0.001, 0.005, None),
    "c5": (6_000_000, 1080, 1920, 8, 0.01, 0.1, 64.0),
}


def config_scene(name):
    n, H, W, seed, lo, hi, zq = CONFIGS[name]
    return synth_scene(n, seed, lo, hi, zq), H, W


def pose_tz(tz):
    p = np.eye(4, dtype=np.float32); p[2, 3] = tz
    return p


def create_orbit_poses(radius, n_views, elevation=30.0, center=None):
    """Look-at orbit, pose[:3,:3] = [right, up, -forward] as columns, pose[:3,3] = camera position
    (visualization.py:156-178).  NOTE the renderer applies this matrix as world->camera without
    inverting it, exactly like the reference (gaussian_splatting.py:250-253)."""
    center = np.zeros(3, np.float32) if center is None else np.asarray(center, np.float32)
    nrm = lambda v: (v / max(float(np.linalg.norm(v)), 1e-12)).astype(np.float32)
    poses = []
    for i in range(n_views):
        ang = 2 * np.pi * i / n_views
        x = radius * np.cos(ang) * np.cos(np.radians(elevation))
        y = radius * np.sin(ang) * np.cos(np.radians(elevation))
        z = radius * np.sin(np.radians(elevation))
        pos = np.array([x, y, z], np.float32)
        fwd = nrm(center - pos)
        up = np.array([0.0, 0.0, 1.0], np.float32)
        right = nrm(np.cross(fwd, up))
        up = nrm(np.cross(right, fwd))
        pose = np.eye(4, dtype=np.float32)
        pose[:3, 0] = right; pose[:3, 1] = up; pose[:3, 2] = -fwd; pose[:3, 3] = pos
        poses.append(pose)
    return np.stack(poses)
