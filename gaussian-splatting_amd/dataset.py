"""the step in front of the render path in training -- a reader for the reference's on-disk scene
format (what /root/reference/data_processing/preprocessor.py:177-220,262-293 writes and
data_processing/dataset.py:11-216 reads):

    <root>/images/<split>/*.jpg          one view per file, name order = pose order
    <root>/poses/<split>_poses.npy       (n, 4, 4) world->camera
    <root>/metadata/scene_info.json

The interface kept from the reference is what its trainer consumes: `SceneDataset(root, split, transform, image_size)`,
`len()`, and items `{'image': (3,H,W) float32 in [0,1], 'camera_pose': (4,4) float32, 'image_path': str}`, with the
same exception types for a bad split, missing files and an image/pose count mismatch.  Everything else is this
package's own: the scene is resolved once into a `SceneFiles` record, views can be fetched as raw uint8 (`view_u8`) so
that the prefetcher moves one byte per channel over PCIe and scales on the GPU.  The reference class's other public
members (`data_path`, `scene_info`, `image_paths`, `camera_poses`, `get_camera_intrinsics()`, `get_scene_bounds()`) are
thin views of the same record, so code written against the reference's dataset keeps working.
The reference decodes with OpenCV, which this image does not have; PIL decodes the same JPEG bytes, and the optional
resize uses PIL's BOX filter where the reference uses cv2.INTER_AREA (both area averages; not bit-identical -- "parity
unpinned" for the resize, no fixture of the reference covers it)."""
import json
import os
from typing import NamedTuple

import numpy as np
import torch
from torch.utils.data import Dataset

SPLITS = ('train', 'val', 'test')


class SceneFiles(NamedTuple):
    views: tuple              # image file paths, name order
    poses: np.ndarray         # (n, 4, 4) float32
    info: dict                # metadata/scene_info.json, parsed


def resolve_scene(root, split) -> SceneFiles:
    """Checks the three parts of a split and loads the small ones (poses, metadata)."""
    if split not in SPLITS:
        raise ValueError(f"unknown split {split!r}: a scene has {', '.join(SPLITS)}")
    root = os.fspath(root)
    parts = {'info': os.path.join(root, 'metadata', 'scene_info.json'),
             'views': os.path.join(root, 'images', split),
             'poses': os.path.join(root, 'poses', split + '_poses.npy')}
    for what, path in parts.items():
        if not os.path.exists(path):
            raise FileNotFoundError(f"scene {root!r}, split {split!r}: no {what} at {path}")
    with open(parts['info']) as f:
        info = json.load(f)
    views = tuple(sorted(os.path.join(parts['views'], name) for name in os.listdir(parts['views']) if name.endswith('.jpg')))
    poses = np.load(parts['poses'], allow_pickle=False)
    if poses.ndim != 3 or poses.shape[1:] != (4, 4) or poses.shape[0] != len(views):
        raise ValueError(f"scene {root!r}, split {split!r}: {len(views)} views but poses of shape {poses.shape}")
    return SceneFiles(views, np.ascontiguousarray(poses, dtype=np.float32), info)


def write_scene(root, splits, metadata=None, quality=95) -> dict:
    """Writes a scene in the layout above, as /root/reference/data_processing/preprocessor.py does for its three splits
    (:177-220 files and poses, :241-247 directories, :262-293 scene_info.json), from views already in memory -- e.g.
    frames of `GaussianSplat.render_frames` -- rather than from a raw capture directory.
    splits: {'train'|'val'|'test': (views, poses)} with views (n, H, W, 3) uint8 or (n, 3, H, W) float in [0, 1] (array or
    tensor) and poses (n, 4, 4).  Splits left out are written empty, so every split of the result resolves.
    Files are named '<split>_<index:06d>.jpg' (name order = pose order), JPEG quality 95 (OpenCV's default; the encoded
    bytes are PIL's, not OpenCV's: "parity unpinned" for them, the layout and the JSON keys are the reference's).
    Returns the scene_info dict."""
    from PIL import Image
    root = os.fspath(root)
    unknown = set(splits) - set(SPLITS)
    if unknown:
        raise ValueError(f"unknown split(s) {sorted(unknown)}: a scene has {', '.join(SPLITS)}")
    sizes, image_size = {}, None
    os.makedirs(os.path.join(root, 'poses'), exist_ok=True)
    os.makedirs(os.path.join(root, 'metadata'), exist_ok=True)
    for split in SPLITS:
        views, poses = splits.get(split, (np.zeros((0, 1, 1, 3), np.uint8), np.zeros((0, 4, 4), np.float32)))
        if torch.is_tensor(views):
            views = views.detach().cpu().numpy()
        if torch.is_tensor(poses):
            poses = poses.detach().cpu().numpy()
        views, poses = np.asarray(views), np.asarray(poses, dtype=np.float32).reshape(-1, 4, 4)
        if views.ndim != 4 or len(views) != len(poses):
            raise ValueError(f"split {split!r}: {views.shape} views but poses of shape {poses.shape}")
        if views.dtype != np.uint8:                                       # (n, 3, H, W) in [0, 1] -> (n, H, W, 3) bytes
            views = np.rint(np.clip(views.transpose(0, 2, 3, 1), 0.0, 1.0) * 255.0).astype(np.uint8)
        if views.shape[-1] != 3:
            raise ValueError(f"split {split!r}: views must be (n, H, W, 3) uint8 or (n, 3, H, W) float, got {views.shape}")
        folder = os.path.join(root, 'images', split)
        os.makedirs(folder, exist_ok=True)
        for i, rgb in enumerate(views):
            Image.fromarray(np.ascontiguousarray(rgb)).save(os.path.join(folder, f"{split}_{i:06d}.jpg"), quality=quality)
        np.save(os.path.join(root, 'poses', split + '_poses.npy'), poses)
        sizes[split] = len(views)
        if len(views):
            image_size = [int(views.shape[2]), int(views.shape[1])]         # (W, H), the reference's order
    info = {'metadata': dict(metadata or {}), 'processing': {'image_size': image_size, 'splits': sizes}}
    with open(os.path.join(root, 'metadata', 'scene_info.json'), 'w') as f:
        json.dump(info, f, indent=4)
    return info


class SceneDataset(Dataset):
    def __init__(self, data_path, split='train', transform=None, image_size=None):
        from pathlib import Path
        self.data_path = Path(data_path)
        self.files = resolve_scene(data_path, split)
        self.split, self.transform = split, transform
        self.image_size = None if image_size is None else (int(image_size[0]), int(image_size[1]))     # (W, H), as for cv2.resize

    info = property(lambda self: self.files.info)

    # ---- the reference class's public surface over the same record (data_processing/dataset.py:34-47,168-216): code
    # written against it reads these attributes and calls the two helpers ------------------------------------------
    @property
    def scene_info(self) -> dict:
        return self.files.info

    @property
    def image_paths(self) -> list:
        """sorted pathlib.Path objects, as `sorted(image_dir.glob('*.jpg'))` gives the reference"""
        from pathlib import Path
        return [Path(v) for v in self.files.views]

    @property
    def camera_poses(self) -> np.ndarray:
        return self.files.poses

    def get_camera_intrinsics(self) -> dict:
        """focal_length, principal_point (cx, cy), image_size (W, H) from the metadata; when the metadata has no camera
        block, the reference's fallback: max(H, W), the image centre and the size of the first view as loaded (:178-193)."""
        try:
            cam = self.files.info['metadata']['camera']
            return {'focal_length': cam['focal_length'], 'principal_point': tuple(cam['principal_point']),
                    'image_size': tuple(self.files.info['processing']['image_size'])}
        except KeyError:
            h, w = self.view_u8(0).shape[:2]
            return {'focal_length': max(h, w), 'principal_point': (w / 2, h / 2), 'image_size': (w, h)}

    def get_scene_bounds(self) -> dict:
        """center (3,), extent (3,), aabb (2, 3) of the camera positions pose[:3, 3] (:195-216)."""
        positions = self.files.poses[:, :3, 3]
        lo, hi = positions.min(axis=0), positions.max(axis=0)
        return {'center': (lo + hi) / 2, 'extent': hi - lo, 'aabb': np.stack([lo, hi])}

    def __len__(self):
        return len(self.files.views)

    def view_u8(self, idx) -> np.ndarray:
        """View idx as (H, W, 3) uint8 RGB, resized to image_size when one was given."""
        from PIL import Image
        path = self.files.views[idx]
        try:
            with Image.open(path) as im:
                im = im.convert('RGB')
                if self.image_size is not None and im.size != self.image_size:
                    im = im.resize(self.image_size, Image.BOX)
                return np.array(im, dtype=np.uint8)
        except (OSError, ValueError) as e:
            raise ValueError(f"cannot decode view {idx} of split {self.split!r}: {path}") from e

    def pose(self, idx) -> torch.Tensor:
        return torch.from_numpy(self.files.poses[idx].copy())

    def __getitem__(self, idx):
        rgb = self.view_u8(idx)
        image = self.transform(rgb) if self.transform is not None else torch.from_numpy(rgb).permute(2, 0, 1).float().div_(255.0)
        return {'image': image, 'camera_pose': self.pose(idx), 'image_path': self.files.views[idx]}


class DevicePrefetcher:
    """Iterates a dataset (or any sequence of SceneDataset-shaped items) and yields (image (1,3,H,W), camera_pose
    (1,4,4), image_path) already on `device`.  Item k+1 is decoded, staged in pinned memory and copied on a side stream
    while the caller renders item k; the caller's stream waits on the copy's event only.
    A SceneDataset without a transform is read as uint8 (`view_u8`): one byte per channel crosses PCIe and the
    conversion to float32 in [0,1] runs on the GPU through a 256-entry table of the host's own i / 255 values, so the
    tensor is bit-identical to the dataset's item."""

    def __init__(self, dataset, device, order=None):
        self.dataset, self.device = dataset, torch.device(device)
        self.order = list(range(len(dataset))) if order is None else list(order)
        self.cuda = self.device.type == 'cuda'
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None
        self.as_bytes = self.cuda and hasattr(dataset, 'view_u8') and getattr(dataset, 'transform', None) is None
        self.unit = (torch.arange(256, dtype=torch.float32) / 255.0).to(self.device) if self.as_bytes else None

    def __len__(self):
        return len(self.order)

    def _stage(self, idx):
        if self.as_bytes:
            raw = torch.from_numpy(self.dataset.view_u8(idx)).pin_memory()                   # (H, W, 3) uint8
            pose = self.dataset.pose(idx).unsqueeze(0).pin_memory()
            path = self.dataset.files.views[idx]
            with torch.cuda.stream(self.stream):
                d_raw = raw.to(self.device, non_blocking=True)
                d_img = self.unit.index_select(0, d_raw.permute(2, 0, 1).reshape(-1).int()).view(1, 3, raw.shape[0], raw.shape[1])
                d_pose = pose.to(self.device, non_blocking=True)
                ev = torch.cuda.Event(); ev.record(self.stream)
            return d_img, d_pose, path, (ev, raw, pose)
        item = self.dataset[idx]
        img, pose = item['image'].unsqueeze(0).contiguous(), item['camera_pose'].unsqueeze(0).contiguous()
        if not self.cuda:
            return img, pose, item['image_path'], None
        img, pose = img.pin_memory(), pose.pin_memory()
        with torch.cuda.stream(self.stream):
            d_img = img.to(self.device, non_blocking=True); d_pose = pose.to(self.device, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(self.stream)
        return d_img, d_pose, item['image_path'], (ev, img, pose)          # keep the pinned buffers alive until the event

    def __iter__(self):
        nxt = self._stage(self.order[0]) if self.order else None
        for k in range(len(self.order)):
            cur = nxt
            nxt = self._stage(self.order[k + 1]) if k + 1 < len(self.order) else None
            if cur[3] is not None:
                torch.cuda.current_stream(self.device).wait_event(cur[3][0])
                cur[0].record_stream(torch.cuda.current_stream(self.device)); cur[1].record_stream(torch.cuda.current_stream(self.device))
            yield cur[0], cur[1], cur[2]
