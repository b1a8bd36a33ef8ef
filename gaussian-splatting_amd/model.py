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
    def render_frames(self, camera_poses: torch.Tensor, image_size: Tuple[int, int], sink=None, block: int = 16):
        """Inference over V poses (V,4,4) -> (V,3,H,W): what the reference's orbit script gets from V calls of
        `model(pose.unsqueeze(0), size)['rendered_images'][0]` (utils/visualization.py:343-347), in one call.
        With `sink` (a frames.FrameSink: animated GIF or a directory of PNGs, visualization.py:180-201,244-266) the views are
        rendered `block` at a time and handed over as they come -- conversion, copy and encoding overlap the next block's
        rendering, and no (V,3,H,W) tensor is ever held; the number of frames written is returned instead."""
        if sink is None:
            return render_orbit(self.positions, self.scales, self.rotations, self.opacity, self.features, camera_poses, image_size)
        V = camera_poses.shape[0]
        for v0 in range(0, V, max(1, int(block))):
            sink.write(render_orbit(self.positions, self.scales, self.rotations, self.opacity, self.features,
                                    camera_poses[v0:v0 + block], image_size))
        return V

    # ---- maintenance of the parameter buffers (gaussian_splatting.py:346-528).  None of the
    # reference's scripts call these; they are here so that a user of the module finds the same methods.  On the
    # GPU prune and clamp run as HIP passes (csrc/gsx_maint.hip); CPU tensors (host-side tooling) use torch ops.
    def _replace_parameters(self, tensors):
        was_flat = self.flat_parameters() is not None
        for k, t in zip(("positions", "scales", "rotations", "opacity", "features"), tensors):
            setattr(self, k, nn.Parameter(t))
        self.num_gaussians = int(tensors[0].shape[0])
        if was_flat:
            self.flatten_parameters()

    # ---- the five parameters as views of ONE buffer.  Same layout as the gradient buffer the render backward
    # writes (rotations | positions | scales | opacity | features: rotations first for their 16-byte loads), so that whatever
    # walks parameters and gradients together -- the broadcast at start-up, an optimizer over slices -- sees two flat arrays.
    FLAT_ORDER = ("rotations", "positions", "scales", "opacity", "features")

    def flat_parameters(self):
        """The 14 N buffer the parameters are consecutive views of, or None (never flattened, or re-seated since -- a
        `.to(device)` moves every parameter on its own)."""
        flat = getattr(self, "_flat", None)
        if flat is None:
            return None
        off = flat.data_ptr()
        for k in self.FLAT_ORDER:
            p = getattr(self, k)
            if p.data_ptr() != off or not p.is_contiguous() or p.dtype != flat.dtype or p.device != flat.device:
                self._flat = None                         # stale (the parameters moved): let the old buffer go
                return None
            off += p.numel() * p.element_size()
        return flat

    @torch.no_grad()
    def flatten_parameters(self):
        """Re-seats the five parameters (the same nn.Parameter objects: optimizers keep working) as views of one contiguous
        buffer and returns it.  Maintenance methods that replace the parameters (prune, add, merge) flatten again."""
        flat = self.flat_parameters()
        if flat is not None:
            return flat
        ps = [getattr(self, k) for k in self.FLAT_ORDER]
        dev, dt = ps[0].device, ps[0].dtype
        if any(p.device != dev or p.dtype != dt for p in ps):
            raise RuntimeError("flatten_parameters: the five parameters must share a device and a dtype")
        flat = torch.empty(sum(p.numel() for p in ps), dtype=dt, device=dev)
        off = 0
        for p in ps:
            view = flat[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            off += p.numel()
        self._flat = flat
        return flat

    @torch.no_grad()
    def prune_gaussians(self, threshold: float = 0.01) -> None:
        """Removes the Gaussians whose opacity is not above `threshold` (:346-369); order of the rest is kept.
        Like the reference this replaces the nn.Parameters (optimizers must be rebuilt)."""
        params = [self.positions, self.scales, self.rotations, self.opacity, self.features]
        if not self.opacity.is_cuda:
            mask = self.opacity.squeeze(-1) > threshold
            self._replace_parameters([p.data[mask] for p in params])
            return
        import ctypes as C
        from . import _lib as L
        lib = L.lib()
        n = self.opacity.shape[0]; dev = self.opacity.device
        ptr = lambda t: C.c_void_p(t.data_ptr())
        keep = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        offsets = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        total = torch.zeros(1, dtype=torch.int64, device=dev)
        wsb = lib.gsx_scan_workspace_bytes(max(n, 1)); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        out = []
        with L.on_device(dev):                       # launches follow the parameters' device, not the current one
            st = L.stream_ptr(dev)
            L.check(lib.gsx_keep_flags(ptr(self.opacity.data.contiguous()), n, float(threshold), ptr(keep), st), "gsx_keep_flags")
            L.check(lib.gsx_scan_u32(ptr(keep), ptr(offsets), ptr(total), n, ptr(ws), wsb, st), "gsx_scan_u32")
            kept = int(total.item())
            for p in params:
                src = p.data.contiguous().float()
                row = src.numel() // max(n, 1)
                dst = torch.empty((kept,) + tuple(src.shape[1:]), dtype=torch.float32, device=dev)
                if kept:
                    L.check(lib.gsx_compact_rows(ptr(src), ptr(keep), ptr(offsets), n, row, ptr(dst), st), "gsx_compact_rows")
                out.append(dst)
        self._replace_parameters(out)

    def add_gaussians(self, num_new: int) -> None:
        """Appends `num_new` Gaussians initialised like the reference (:371-402): positions ~ N(0,1), unit scales,
        identity rotations, zero opacity, colours ~ U(0,1) -- same draws in the same order on the parameters' device."""
        dev = self.positions.device
        new_positions = torch.randn(num_new, 3, device=dev)
        new_scales = torch.ones(num_new, 3, device=dev)
        new_rotations = torch.zeros(num_new, 4, device=dev)
        new_opacity = torch.zeros(num_new, 1, device=dev)
        new_features = torch.rand(num_new, 3, device=dev)
        new_rotations[:, 0] = 1
        with torch.no_grad():
            self._replace_parameters([torch.cat([self.positions.data, new_positions]), torch.cat([self.scales.data, new_scales]),
                                      torch.cat([self.rotations.data, new_rotations]), torch.cat([self.opacity.data, new_opacity]),
                                      torch.cat([self.features.data, new_features])])

    @torch.no_grad()
    def optimize_scales(self) -> None:
        """Clamps the scales to [1e-4, 1.0] (:404-418), in place."""
        if not self.scales.is_cuda:
            self.scales.data = torch.minimum(torch.maximum(self.scales.data, torch.tensor(0.0001)), torch.tensor(1.0))
            return
        import ctypes as C
        from . import _lib as L
        s = self.scales.data
        if not s.is_contiguous():
            s = s.contiguous(); self.scales.data = s
        with L.on_device(s.device):
            L.check(L.lib().gsx_clamp_inplace(C.c_void_p(s.data_ptr()), s.numel(), 0.0001, 1.0,
                                              L.stream_ptr(s.device)), "gsx_clamp_inplace")
        torch.autograd.graph.increment_version(self.scales)      # written through a raw pointer: autograd must see it

    @torch.no_grad()
    def get_dense_points(self, num_points: int = 1000000) -> torch.Tensor:
        """(num_points, 6) xyz+rgb samples, Gaussians drawn with probability softmax(opacity) (:420-443)."""
        weights = F.softmax(self.opacity.squeeze(-1), dim=0)
        indices = torch.multinomial(weights, num_points, replacement=True)
        return torch.cat([self.positions[indices], self.features[indices]], dim=1)

    def save_state(self, path: str) -> None:
        """Same file layout as the reference (:445-461): a dict of num_gaussians and the five tensors."""
        torch.save({'num_gaussians': self.num_gaussians, 'positions': self.positions.data, 'scales': self.scales.data,
                    'rotations': self.rotations.data, 'opacity': self.opacity.data, 'features': self.features.data}, path)

    def load_state(self, path: str) -> None:
        """:463-476.  Loaded with weights_only=True (tensors and an int: nothing in the file is executed)."""
        state = torch.load(path, weights_only=True)
        self._replace_parameters([state[k] for k in ("positions", "scales", "rotations", "opacity", "features")])
        self.num_gaussians = state['num_gaussians']

    @torch.no_grad()
    def merge_nearby_gaussians(self, distance_threshold: float = 0.01) -> None:
        """Greedy merge of pairs closer than the threshold (:478-528): pairs (i < j) in row-major order, a Gaussian
        takes part in at most one merge as the removed one, the survivor gets the opacity-weighted mean of position
        and colour (of the CURRENT values, as in the reference) and the larger opacity.
        The reference materialises the N x N matrix of torch.cdist (4 TB at 1 M Gaussians).  On the GPU the pairs come
        from a spatial hash (csrc/gsx_merge.hip: O(N) memory, direct fp32 distances), the order-dependent walk runs on
        the host inside the library and the survivors are compacted on the GPU.  CPU tensors (host-side tooling) scan
        the same matrix in row blocks with torch.cdist itself."""
        if self.positions.is_cuda:
            return self._merge_nearby_hip(float(distance_threshold))
        n = self.num_gaussians
        pos0 = self.positions.data
        pairs = []
        block = max(1, min(n, (1 << 24) // max(n, 1)))
        for i0 in range(0, n, block):
            d = torch.cdist(pos0[i0:i0 + block], pos0)
            ii, jj = torch.where((d > 0) & (d < distance_threshold))
            ii = ii + i0
            sel = jj >= ii         # torch.triu keeps the diagonal: a self-distance that rounds above 0 "merges" i with itself (and removes it), as in the reference
            pairs.append(torch.stack([ii[sel], jj[sel]], 1).cpu())
        pairs = torch.cat(pairs) if pairs else torch.zeros(0, 2, dtype=torch.long)
        if pairs.shape[0] == 0:
            return
        # the order-dependent walk over the pair list is the library's host routine (the one the GPU path uses): the list
        # as CSR rows (row-major order == ascending i, then j), host copies of the three tensors it rewrites
        import ctypes as C
        from . import _lib as L
        ptr = lambda t: C.c_void_p(t.data_ptr())
        row_start = torch.zeros(n + 1, dtype=torch.int32)
        row_start[1:] = torch.cumsum(torch.bincount(pairs[:, 0], minlength=n), 0).to(torch.int32)
        partners = pairs[:, 1].to(torch.int32).contiguous()
        host = [t.data.float().cpu().contiguous().clone() for t in (self.positions, self.features, self.opacity)]
        removed = torch.empty(n, dtype=torch.uint8)
        L.check(L.lib().gsx_merge_greedy_host(n, ptr(row_start), ptr(partners), ptr(host[0]), ptr(host[1]), ptr(host[2]), ptr(removed)),
                "gsx_merge_greedy_host")
        keep = removed == 0
        if bool(keep.all()):
            kept = [host[0], self.scales.data, self.rotations.data, host[2], host[1]]
        else:
            kept = [host[0][keep], self.scales.data[keep], self.rotations.data[keep], host[2][keep], host[1][keep]]
        self._replace_parameters([t.to(pos0.device) for t in kept])

    def _merge_pairs_hip(self, threshold: float):
        """(offsets (N+1,) int64 host, partners (P,) int32 host): i's partners j > i, ascending, for every i."""
        import ctypes as C
        from . import _lib as L
        lib = L.lib()
        n = self.num_gaussians; dev = self.positions.device
        ptr = lambda t: C.c_void_p(t.data_ptr())
        pos = self.positions.data.contiguous().float()
        counts = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        offsets = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        total = torch.zeros(2, dtype=torch.int64, device=dev)
        wsb = lib.gsx_merge_workspace_bytes(n); ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        swb = lib.gsx_scan_workspace_bytes(max(n, 1)); sws = torch.empty(swb, dtype=torch.uint8, device=dev)
        with L.on_device(dev):
            st = L.stream_ptr(dev)
            L.check(lib.gsx_merge_count(ptr(pos), n, threshold, ptr(counts), ptr(ws), wsb, st), "gsx_merge_count")
            L.check(lib.gsx_scan_u32(ptr(counts), ptr(offsets), ptr(total), n, ptr(sws), swb, st), "gsx_scan_u32")
            n_pairs = int(total[0].item())
            if n_pairs >= 1 << 31:
                raise ValueError(f"merge_nearby_gaussians: {n_pairs} pairs closer than {threshold}; lower the threshold")
            partners = torch.empty(max(n_pairs, 1), dtype=torch.int32, device=dev)
            if n_pairs:
                L.check(lib.gsx_merge_fill(ptr(pos), n, threshold, ptr(offsets), ptr(partners), ptr(ws), wsb, st), "gsx_merge_fill")
            off_host = torch.empty(n + 1, dtype=torch.int32)
            off_host[:n] = offsets[:n].cpu(); off_host[n] = n_pairs
            return off_host, partners[:n_pairs].cpu()

    def _merge_nearby_hip(self, threshold: float) -> None:
        import ctypes as C
        from . import _lib as L
        lib = L.lib()
        n = self.num_gaussians; dev = self.positions.device
        if n == 0 or not threshold > 0 or threshold == float("inf"):     # no d satisfies 0 < d < threshold (or the N^2 pairs of inf: refuse)
            if threshold == float("inf"):
                raise ValueError("merge_nearby_gaussians: an infinite threshold pairs every two Gaussians")
            return
        ptr = lambda t: C.c_void_p(t.data_ptr())
        off_host, partners_host = self._merge_pairs_hip(threshold)
        if partners_host.numel() == 0:
            return
        pos_h = self.positions.data.float().cpu().contiguous()
        fea_h = self.features.data.float().cpu().contiguous()
        opa_h = self.opacity.data.float().cpu().contiguous()
        removed = torch.empty(n, dtype=torch.uint8)
        L.check(lib.gsx_merge_greedy_host(n, ptr(off_host), ptr(partners_host), ptr(pos_h), ptr(fea_h), ptr(opa_h), ptr(removed)),
                "gsx_merge_greedy_host")
        keep = (removed == 0).to(torch.int32).to(dev)
        srcs = [pos_h.to(dev), self.scales.data.contiguous().float(), self.rotations.data.contiguous().float(), opa_h.to(dev), fea_h.to(dev)]
        offsets = torch.empty(n, dtype=torch.int32, device=dev)
        total = torch.zeros(2, dtype=torch.int64, device=dev)
        swb = lib.gsx_scan_workspace_bytes(n); sws = torch.empty(swb, dtype=torch.uint8, device=dev)
        out = []
        with L.on_device(dev):
            st = L.stream_ptr(dev)
            L.check(lib.gsx_scan_u32(ptr(keep), ptr(offsets), ptr(total), n, ptr(sws), swb, st), "gsx_scan_u32")
            kept = int(total[0].item())
            for src in srcs:
                dst = torch.empty((kept,) + tuple(src.shape[1:]), dtype=torch.float32, device=dev)
                if kept:
                    L.check(lib.gsx_compact_rows(ptr(src), ptr(keep), ptr(offsets), n, src.numel() // n, ptr(dst), st), "gsx_compact_rows")
                out.append(dst)
        self._replace_parameters(out)

    def last_n_composited(self):
        """Number of Gaussians composited before the global early exit (:147) in the last
        forward, per view (synchronises)."""
        if self._last_state is None:
            return None
        s = self._last_state.cpu().numpy().astype('uint32')
        n = self.num_gaussians
        return [int(v) + 1 if (v != 0xFFFFFFFF and int(v) + 1 < n) else n for v in s]
