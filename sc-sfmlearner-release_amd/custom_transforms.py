"""Joint image / intrinsics transforms with the reference's names and behaviour
(custom_transforms.py): lists of HxWx3 float arrays + a 3x3 intrinsics matrix in, the same out."""
from __future__ import division

import random

import numpy as np
import torch
from PIL import Image


class Compose(object):
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, images, intrinsics):
        for t in self.transforms:
            images, intrinsics = t(images, intrinsics)
        return images, intrinsics


class Normalize(object):
    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, images, intrinsics):
        m = torch.tensor(self.mean, dtype=torch.float32).view(-1, 1, 1)
        s = torch.tensor(self.std, dtype=torch.float32).view(-1, 1, 1)
        for t in images:
            t.sub_(m).div_(s)
        return images, intrinsics


class ArrayToTensor(object):
    """HxWxC arrays in [0,255] -> CxHxW float tensors in [0,1]."""

    def __call__(self, images, intrinsics):
        return [torch.from_numpy(np.ascontiguousarray(np.transpose(im, (2, 0, 1)))).float() / 255 for im in images], intrinsics


class RandomHorizontalFlip(object):
    """Flip with probability 0.5; the principal point follows (custom_transforms.py:46-59)."""

    def __call__(self, images, intrinsics):
        assert intrinsics is not None
        if random.random() < 0.5:
            out_k = np.copy(intrinsics)
            out = [np.copy(np.fliplr(im)) for im in images]
            out_k[0, 2] = out[0].shape[1] - out_k[0, 2]
            return out, out_k
        return images, intrinsics


class RandomScaleCrop(object):
    """Zoom by up to 15 % per axis and crop back to the input size (custom_transforms.py:62-84)."""

    def __call__(self, images, intrinsics):
        assert intrinsics is not None
        out_k = np.copy(intrinsics)
        in_h, in_w, _ = images[0].shape
        x_s, y_s = np.random.uniform(1, 1.15, 2)
        sh, sw = int(in_h * y_s), int(in_w * x_s)
        out_k[0] *= x_s
        out_k[1] *= y_s
        scaled = [np.array(Image.fromarray(im.astype(np.uint8)).resize((sw, sh))).astype(np.float32) for im in images]
        oy = np.random.randint(sh - in_h + 1)
        ox = np.random.randint(sw - in_w + 1)
        out_k[0, 2] -= ox
        out_k[1, 2] -= oy
        return [im[oy:oy + in_h, ox:ox + in_w] for im in scaled], out_k


class ArrayToUint8(object):
    """Extension for the device-side transform (scsfm_hip/augment.py): keep the decoded frames as
    HxWx3 uint8 tensors; flip / zoom-crop / normalisation then run on the GPU."""

    def __call__(self, images, intrinsics):
        return [torch.from_numpy(np.ascontiguousarray(im).astype(np.uint8)) for im in images], intrinsics
