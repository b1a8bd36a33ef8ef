#!/usr/bin/env python3
"""Builds tests/golden/scene_n2/: a small processed scene in the reference's on-disk layout
(/root/reference/data_processing/dataset.py:97-120 reads it, preprocessor.py:177-220,262-293 writes it) that this package's
own writer (dataset.write_scene) did NOT produce -- so the reader is tested on files it has not seen the making of:
views of odd sizes (17x11, 9x13), one GRAYSCALE JPEG, one progressive JPEG, names that sort differently as strings than
as numbers, a stray non-.jpg file and a sub-directory inside the split folder, float64 poses, metadata WITHOUT a camera
block.  The source rasters are kept next to the files (rasters.npz) so the test can bound the JPEG error.
Decode / resize parity with the reference stays UNPINNED: the reference decodes with OpenCV (absent in this image) and no
file of the reference holds decoded pixels; PIL both encodes (here) and decodes (in the reader).
Run once; the bytes are committed.  usage: python tests/golden/make_scene_n2.py"""
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scene_n2")
rng = np.random.default_rng(20261111)


def smooth(h, w, c):
    """a low-frequency raster (JPEG keeps it within a few grey levels)"""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    planes = [127 + 100 * np.sin(x / (3 + k) + k) * np.cos(y / (4 + k)) for k in range(c)]
    return np.clip(np.stack(planes, -1), 0, 255).astype(np.uint8)


os.makedirs(os.path.join(HERE, "images", "train", "thumbs"), exist_ok=True)
os.makedirs(os.path.join(HERE, "images", "val"), exist_ok=True)
os.makedirs(os.path.join(HERE, "poses"), exist_ok=True)
os.makedirs(os.path.join(HERE, "metadata"), exist_ok=True)
rasters = {}
# name order (what the reference's sorted(glob('*.jpg')) gives): view_10 < view_2 < view_3 < view_a
train = [("view_10.jpg", smooth(11, 17, 3), {}), ("view_2.jpg", smooth(11, 17, 1)[..., 0], {}),          # grayscale file
         ("view_3.jpg", smooth(11, 17, 3)[::-1].copy(), {"progressive": True}), ("view_a.jpg", smooth(11, 17, 3)[:, ::-1].copy(), {"quality": 60})]
for name, arr, kw in train:
    Image.fromarray(arr).save(os.path.join(HERE, "images", "train", name), quality=kw.pop("quality", 92), **kw)
    rasters["train/" + name] = arr
Image.fromarray(smooth(13, 9, 3)).save(os.path.join(HERE, "images", "val", "0.jpg"), quality=92)
rasters["val/0.jpg"] = smooth(13, 9, 3)
with open(os.path.join(HERE, "images", "train", "README.txt"), "w") as f:
    f.write("not a view\n")
Image.fromarray(smooth(4, 4, 3)).save(os.path.join(HERE, "images", "train", "view_1.png"))                 # not a .jpg: ignored
Image.fromarray(smooth(4, 4, 3)).save(os.path.join(HERE, "images", "train", "thumbs", "t.jpg"))           # glob is not recursive
poses = np.tile(np.eye(4), (4, 1, 1))
poses[:, :3, 3] = rng.uniform(-2, 2, (4, 3))
np.save(os.path.join(HERE, "poses", "train_poses.npy"), poses)                                            # float64, as np.stack of loadtxt gives
np.save(os.path.join(HERE, "poses", "val_poses.npy"), np.eye(4)[None].astype(np.float32))
with open(os.path.join(HERE, "metadata", "scene_info.json"), "w") as f:
    json.dump({"metadata": {"scene_name": "n2_fixture"}, "processing": {"image_size": [17, 11], "splits": {"train": 4, "val": 1, "test": 0}}}, f, indent=4)
np.savez(os.path.join(HERE, "rasters.npz"), **{k.replace("/", "__"): v for k, v in rasters.items()})
print("wrote", HERE)
