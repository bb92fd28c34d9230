"""Blender-synthetic scene ingest (reference: nerf/dataset.py:34-135).

Same class name, constructor arguments, item format and camera accessors as the reference's ``CustomDataSet`` so that its
entry scripts can import it, but written against PIL + numpy only (torchvision / natsort are not required; a torchvision
``transforms.Compose`` passed as ``transform`` is still honoured).  Host-side code: nothing here runs on the GPU.

Deliberate deviation: the reference's file filter keeps ``*_depth_*.png`` in ``test/`` (dataset.py:45),
which shifts every image against its pose; depth maps are excluded here as well.
"""
import json
import os
import re
from typing import Callable, Optional

import numpy as np
import torch
from PIL import Image

DATASET_PREFIX = "../../dataset/nerf_synthetic/"


def _natural_key(name: str):
    return [int(tok) if tok.isdigit() else tok.lower() for tok in re.split(r"(\d+)", name)]


def to_tensor(image: Image.Image) -> torch.Tensor:
    """PIL image -> float tensor (C, H, W) in [0, 1] (what torchvision's ToTensor does for 8-bit images)."""
    arr = np.asarray(image, dtype=np.uint8)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1))).float().div_(255.0)


class AdaptiveResize(torch.nn.Module):
    """Bilinear resize by a ratio (dataset.py:22-33): new size = (int(H * ratio), int(W * ratio))."""

    def __init__(self, ratio):
        super().__init__()
        self.ratio = ratio

    def forward(self, input: Image.Image) -> Image.Image:
        if self.ratio == 1.0:
            return input
        w, h = input.size
        return input.resize((int(w * self.ratio), int(h * self.ratio)), resample=Image.BILINEAR)


class CustomDataSet(torch.utils.data.Dataset):
    def __init__(self, root_dir, transform: Optional[Callable], scene_scale=1.0, is_train=True, use_alpha=False, white_bkg=False,
                 use_div=False):
        self.is_train = is_train
        self.root_dir = root_dir
        self.main_dir = os.path.join(root_dir, "train" if is_train else "test") + "/"
        self.transform = transform if transform is not None else to_tensor
        names = [n for n in os.listdir(self.main_dir) if n.endswith("png") and "normal" not in n and "alpha" not in n and "depth" not in n]
        self.total_imgs = sorted(names, key=_natural_key)
        self.use_alpha = use_alpha
        self.scene_scale = scene_scale
        self.white_bkg = white_bkg
        self.use_div = use_div
        stem = os.path.join(root_dir, "transforms_%s" % ("train" if is_train else "test"))
        self.cam_fov, self.tfs, self.divisions, self.weights = CustomDataSet.readFromJson(stem + ("_div.json" if use_div else ".json"), use_div)

    def __len__(self):
        return len(self.total_imgs)

    def _load(self, idx, rgba: bool) -> torch.Tensor:
        image = Image.open(os.path.join(self.main_dir, self.total_imgs[idx]), mode="r").convert("RGBA" if rgba else "RGB")
        out = self.transform(image)
        return out if isinstance(out, torch.Tensor) else to_tensor(out)

    def __getitem__(self, idx):
        img = self._load(idx, self.use_alpha or self.white_bkg)
        tf = self.tfs[idx].clone()
        if self.white_bkg:                                                     # composite on white with the alpha channel (dataset.py:62-63)
            img = img[:3, ...] * img[-1:, ...] + (1.0 - img[-1:, ...])
        tf[:3, -1] *= self.scene_scale
        return img, tf

    def r_c(self):
        image, _ = self[0]
        return image.shape[1], image.shape[2]

    def cuda(self, flag=True):
        self.is_cuda = flag

    @staticmethod
    def readFromJson(path: str, use_div=False):
        with open(path, "r") as f:
            items = json.load(f)
        cam_fov = items["camera_angle_x"]
        if "camera_angle_y" in items:
            cam_fov = (cam_fov, items["camera_angle_y"])
        tfs = torch.from_numpy(np.stack([np.asarray(fr["transform_matrix"], dtype=np.float64) for fr in items["frames"]], 0))[:, :3, :]
        division = items.get("division", None) if use_div else None
        weights = items.get("weights", None) if use_div else None
        return cam_fov, tfs.float(), division, weights

    def getCameraParam(self):
        return self.cam_fov, self.tfs

    def get_dataset(self, to_cuda: bool):
        """camera fov, per-image transforms, all images stacked (N, C, H, W) -- resident in HBM when to_cuda (288 GB: a whole
        scene fits, so the per-iteration host->device image copy of train.py:153-157 is unnecessary)."""
        imgs = torch.stack([self._load(i, self.use_alpha) for i in range(len(self))], 0).float()
        cam_fov, tfs = self.getCameraParam()
        if to_cuda:
            return cam_fov, tfs.cuda(), imgs.cuda()
        return cam_fov, tfs, imgs


def save_image(tensors, path: str, nrow: int = 8, padding: int = 2):
    """Grid PNG writer with torchvision.utils.save_image's layout (used by render_only): `tensors` = list of (3, H, W) float
    images in [0, 1] (or one (N, 3, H, W) tensor), `nrow` images per row, `padding` black pixels between them."""
    if isinstance(tensors, torch.Tensor):
        tensors = list(tensors) if tensors.dim() == 4 else [tensors]
    imgs = [t.detach().float().cpu().clamp(0, 1) for t in tensors]
    n = len(imgs)
    c, h, w = imgs[0].shape
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    if n == 1:
        grid = imgs[0]
    else:
        grid = torch.zeros(c, rows * (h + padding) + padding, cols * (w + padding) + padding)
        for k, im in enumerate(imgs):
            r, q = divmod(k, cols)
            grid[:, padding + r * (h + padding): padding + r * (h + padding) + h, padding + q * (w + padding): padding + q * (w + padding) + w] = im
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(arr if arr.shape[2] == 3 else arr[:, :, 0]).save(path)
