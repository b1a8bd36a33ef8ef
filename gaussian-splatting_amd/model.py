"""Drop-in `GaussianSplat` module: same constructor, parameter names/shapes/init, forward
signature and output dict as /root/reference/model/gaussian_splatting.py:7-219, rendering
through the gfx950 HIP path instead of the per-Gaussian PyTorch loop.
"""
from typing import Dict, Tuple
import torch
import torch.nn as nn
import torch.nn.functional as F
from .render import render_views, render_orbit


class GaussianSplat(nn.Module):
    """3D-Gaussian scene; parameters `positions (N,3)`, `scales (N,3)`, `rotations (N,4)` (w,x,y,z),
    `opacity (N,1)`, `features (N,3)` -- state_dict keys identical to the reference (:37-41).

    `use_gradient_checkpointing`, `memory_efficient` and `chunk_size` are accepted for drop-in
    compatibility; they are workarounds for the reference's dense loop (autograd keeps ~28 B per
    pixel-Gaussian pair) and have nothing to act on here: the fused kernels keep 52 B per
    tile-instance.
    """

    def __init__(self, num_gaussians=100000, use_gradient_checkpointing=False, memory_efficient=False, chunk_size=500):
        super().__init__()
        self.num_gaussians = num_gaussians
        self.use_gradient_checkpointing = use_gradient_checkpointing
        self.memory_efficient = memory_efficient
        self.chunk_size = chunk_size
        # same draws in the same order as the reference (:37-41), so that a seeded construction
        # yields identical tensors
        self.positions = nn.Parameter(torch.randn(num_gaussians, 3))
        self.scales = nn.Parameter(torch.ones(num_gaussians, 3))
        self.rotations = nn.Parameter(torch.randn(num_gaussians, 4))
        self.opacity = nn.Parameter(torch.ones(num_gaussians, 1))
        self.features = nn.Parameter(torch.randn(num_gaussians, 3))
        self._initialize_parameters()
        self._last_state = None
        # extra output beyond the reference's six keys: 'depth_map' (alpha-composited camera depth); off by default,
        # it costs an accumulator per pixel in the compositing kernels
        self.render_depth_map = False

    def _initialize_parameters(self):
        """:46-63"""
        nn.init.uniform_(self.positions, -1, 1)
        nn.init.uniform_(self.scales, 0.001, 0.1)
        self.rotations.data[:, 0] = 1
        self.rotations.data[:, 1:] = 0.01 * torch.randn_like(self.rotations.data[:, 1:])
        self._normalize_quaternions()
        nn.init.uniform_(self.opacity, 0, 1)
        nn.init.uniform_(self.features, 0, 1)

    def _normalize_quaternions(self):
        """:65-68"""
        with torch.no_grad():
            self.rotations.data = F.normalize(self.rotations.data, dim=1)

    def forward(self, camera_poses: torch.Tensor, image_size: Tuple[int, int]) -> Dict[str, torch.Tensor]:
        """camera_poses (B,4,4) applied as world->camera exactly like the reference (:250-253);
        image_size (H, W).  Returns the reference's six keys (:212-219) plus `alpha (B,1,H,W)`
        (accumulated alpha, which the reference computes but never returns, :97) and, with
        `self.render_depth_map = True`, `depth_map (B,1,H,W)` (alpha-composited camera depth)."""
        if camera_poses.dim() != 3 or camera_poses.shape[-2:] != (4, 4):
            raise ValueError(f"camera_poses must be (B,4,4), got {tuple(camera_poses.shape)}")
        H, W = int(image_size[0]), int(image_size[1])
        rgb, mean2d, depth, cov2d, alpha, dmap, state = render_views(
            self.positions, self.scales, self.rotations, self.opacity, self.features,
            camera_poses.to(self.positions.device), (H, W), want_depth=self.render_depth_map)
        self._last_state = state
        out = {
            'rendered_images': rgb,
            'positions': mean2d,
            'depths': depth,
            'scales': cov2d,
            'features': self.features,
            'opacity': self.opacity,
            'alpha': alpha,
        }
        if self.render_depth_map:
            out['depth_map'] = dmap
        return out

    @torch.no_grad()
    def render_frames(self, camera_poses: torch.Tensor, image_size: Tuple[int, int]) -> torch.Tensor:
        """Inference over V poses (V,4,4) -> (V,3,H,W): what the reference's orbit script gets from V calls of
        `model(pose.unsqueeze(0), size)['rendered_images'][0]` (utils/visualization.py:343-347), in one call."""
        return render_orbit(self.positions, self.scales, self.rotations, self.opacity, self.features, camera_poses, image_size)

    def last_n_composited(self):
        """Number of Gaussians composited before the global early exit (:147) in the last
        forward, per view (synchronises)."""
        if self._last_state is None:
            return None
        s = self._last_state.cpu().numpy().astype('uint32')
        n = self.num_gaussians
        return [int(v) + 1 if (v != 0xFFFFFFFF and int(v) + 1 < n) else n for v in s]
