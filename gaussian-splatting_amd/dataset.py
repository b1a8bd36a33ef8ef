"""the step in front of the render path in training -- the reference's on-disk scene format
(/root/reference/data_processing/dataset.py:11-216, written by data_processing/preprocessor.py:177-220,262-293):

    <root>/images/<split>/*.jpg          sorted by name
    <root>/poses/<split>_poses.npy       (n, 4, 4)
    <root>/metadata/scene_info.json

`SceneDataset` mirrors the reference class (same constructor, items, errors, helper methods).  The reference
decodes with OpenCV, which this image does not have; PIL decodes the same JPEG bytes, and the optional resize uses
PIL's BOX filter where the reference uses cv2.INTER_AREA (both area averages; not bit-identical -- "parity
unpinned" for the resize, no fixture of the reference covers it).  `DevicePrefetcher` is the MI355X side: items are
staged in pinned host memory and copied to the GPU on a side stream one step ahead, so the trainer's stream never
waits on PCIe."""
import json
import logging
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset


class SceneDataset(Dataset):
    def __init__(self, data_path, split='train', transform=None, image_size=None):
        self.data_path = Path(data_path)
        if split not in ['train', 'val', 'test']:
            raise ValueError(f"Split must be one of ['train', 'val', 'test'], got {split}")
        self.split = split
        self.transform = transform
        self.image_size = image_size
        self.logger = logging.getLogger(__name__)
        self.scene_info = self._load_scene_info()
        self.image_paths = self._get_image_paths()
        self.camera_poses = self._load_camera_poses()
        if len(self.image_paths) != len(self.camera_poses):
            raise ValueError(f"Number of images ({len(self.image_paths)}) does not match "
                             f"number of camera poses ({len(self.camera_poses)})")

    def __len__(self):
        return len(self.image_paths)

    def __getitem__(self, idx):
        image_path = self.image_paths[idx]
        image = self._load_image(image_path)
        image = self.transform(image) if self.transform is not None else self._default_transform(image)
        return {'image': image, 'camera_pose': torch.from_numpy(self.camera_poses[idx]).float(), 'image_path': str(image_path)}

    def _load_scene_info(self):
        info_path = self.data_path / 'metadata' / 'scene_info.json'
        if not info_path.exists():
            raise FileNotFoundError(f"Scene info not found at {info_path}")
        with open(info_path) as f:
            return json.load(f)

    def _get_image_paths(self):
        image_dir = self.data_path / 'images' / self.split
        if not image_dir.exists():
            raise FileNotFoundError(f"Image directory not found at {image_dir}")
        return sorted(image_dir.glob('*.jpg'))

    def _load_camera_poses(self):
        poses_path = self.data_path / 'poses' / f'{self.split}_poses.npy'
        if not poses_path.exists():
            raise FileNotFoundError(f"Camera poses not found at {poses_path}")
        return np.load(poses_path)            # allow_pickle stays False

    def _load_image(self, image_path):
        """(H, W, 3) uint8 RGB; image_size, when given, is (W, H) as for cv2.resize (dataset.py:146)."""
        from PIL import Image
        try:
            with Image.open(str(image_path)) as im:
                im = im.convert('RGB')
                if self.image_size is not None:
                    im = im.resize(tuple(self.image_size), Image.BOX)
                return np.asarray(im, dtype=np.uint8).copy()
        except (OSError, ValueError) as e:
            raise ValueError(f"Failed to load image: {image_path}") from e

    def _default_transform(self, image):
        return torch.from_numpy(image.astype(np.float32) / 255.0).permute(2, 0, 1)

    def get_camera_intrinsics(self):
        try:
            camera_info = self.scene_info['metadata']['camera']
            return {'focal_length': camera_info['focal_length'], 'principal_point': tuple(camera_info['principal_point']),
                    'image_size': tuple(self.scene_info['processing']['image_size'])}
        except KeyError:
            self.logger.warning("Camera intrinsics not found in metadata")
            H, W = self._load_image(self.image_paths[0]).shape[:2]
            return {'focal_length': max(H, W), 'principal_point': (W / 2, H / 2), 'image_size': (W, H)}

    def get_scene_bounds(self):
        positions = self.camera_poses[:, :3, 3]
        lo, hi = positions.min(axis=0), positions.max(axis=0)
        return {'center': (lo + hi) / 2, 'extent': hi - lo, 'aabb': np.stack([lo, hi])}


class DevicePrefetcher:
    """Iterates a dataset (or any iterable of SceneDataset items) and yields (image (1,3,H,W), camera_pose (1,4,4),
    image_path) already on `device`.  Item k+1 is decoded, staged in pinned memory and copied on a side stream while
    the caller renders item k; the caller's stream waits on the copy's event only."""

    def __init__(self, dataset, device, order=None):
        self.dataset, self.device = dataset, torch.device(device)
        self.order = list(range(len(dataset))) if order is None else list(order)
        self.cuda = self.device.type == 'cuda'
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None

    def __len__(self):
        return len(self.order)

    def _stage(self, idx):
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
