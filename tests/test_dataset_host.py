"""Host-side ingest / output helpers: Blender-format loader, PNG grid writer."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from nerf_amd.dataset import AdaptiveResize, CustomDataSet, save_image, to_tensor


def make_scene(root, split="train", names=("r_0", "r_1", "r_2", "r_10"), size=(8, 6), seed=0):
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, split), exist_ok=True)
    frames, pixels = [], {}
    for k, n in enumerate(names):
        arr = rng.integers(0, 256, size=(size[0], size[1], 4), dtype=np.uint8)
        Image.fromarray(arr, "RGBA").save(os.path.join(root, split, n + ".png"))
        pixels[n] = arr
        tf = np.eye(4)
        tf[:3, 3] = [k, 2 * k, -k]
        frames.append({"file_path": "./%s/%s" % (split, n), "transform_matrix": tf.tolist()})
    for decoy in ("r_0_depth_0001.png", "r_0_normal_0001.png", "r_1_alpha.png"):
        Image.fromarray(np.zeros((size[0], size[1], 3), np.uint8)).save(os.path.join(root, split, decoy))
    json.dump({"camera_angle_x": 0.69, "frames": frames}, open(os.path.join(root, "transforms_%s.json" % split), "w"))
    return pixels


def test_loader_order_shapes_and_compositing(tmp_path):
    root = str(tmp_path) + "/"
    px = make_scene(root)
    ds = CustomDataSet(root, None, scene_scale=0.5, is_train=True, white_bkg=True)
    assert len(ds) == 4 and ds.total_imgs == ["r_0.png", "r_1.png", "r_2.png", "r_10.png"]      # natural order, decoys skipped
    fov, tfs = ds.getCameraParam()
    assert fov == 0.69 and tfs.shape == (4, 3, 4) and tfs.dtype == torch.float32
    img, tf = ds[3]
    a = torch.from_numpy(px["r_10"].astype(np.float32) / 255.0).permute(2, 0, 1)
    want = a[:3] * a[3:] + (1.0 - a[3:])
    assert img.shape == (3, 8, 6) and torch.allclose(img, want, atol=1e-6)
    assert torch.equal(tf[:, 3], torch.tensor([3.0, 6.0, -3.0]) * 0.5) and torch.equal(ds.tfs[3][:, 3], torch.tensor([3.0, 6.0, -3.0]))
    assert ds.r_c() == (8, 6)
    rgb = CustomDataSet(root, None, is_train=True)[0][0]
    assert rgb.shape == (3, 8, 6) and torch.allclose(rgb, a.new_tensor(px["r_0"][..., :3].astype(np.float32) / 255.0).permute(2, 0, 1), atol=1e-6)
    fov2, tfs2, imgs = CustomDataSet(root, None, is_train=True, use_alpha=True).get_dataset(False)
    assert imgs.shape == (4, 4, 8, 6) and fov2 == 0.69


def test_resize_tuple_fov_and_grid_writer(tmp_path):
    root = str(tmp_path) + "/"
    make_scene(root, split="test", names=("r_0", "r_1"), size=(10, 20))
    items = json.load(open(root + "transforms_test.json"))
    items["camera_angle_y"] = 0.5
    json.dump(items, open(root + "transforms_test.json", "w"))
    resize = AdaptiveResize(0.5)
    ds = CustomDataSet(root, lambda im: to_tensor(resize(im)), is_train=False)
    assert ds.r_c() == (5, 10) and ds.cam_fov == (0.69, 0.5)
    out = os.path.join(root, "out", "grid.png")
    save_image([torch.rand(3, 5, 10), torch.rand(3, 5, 10), torch.rand(3, 5, 10)], out, nrow=2)
    assert Image.open(out).size == (2 * 12 + 2, 2 * 7 + 2)
    save_image([torch.full((3, 4, 4), 0.5)], out, nrow=1)
    assert np.asarray(Image.open(out))[0, 0, 0] == 128
