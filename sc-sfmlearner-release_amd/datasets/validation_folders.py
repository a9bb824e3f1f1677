"""ValidationSet (datasets/validation_folders.py): frames with .npy ground-truth depth next to them;
a sample is (img, depth).  NYU stores depth in millimetres (divided by 1000 on load, :49-52)."""
import os

import numpy as np
import torch
import torch.utils.data as data

from .sequence_folders import load_as_float


class ValidationSet(data.Dataset):
    def __init__(self, root, transform=None, dataset='nyu'):
        self.root = str(root)
        scene_list = os.path.join(self.root, 'val.txt')
        self.scenes = [os.path.join(self.root, line.strip()) for line in open(scene_list) if line.strip()]
        self.transform = transform
        self.dataset = dataset
        self.imgs, self.depth = [], []
        for scene in self.scenes:
            files = sorted(f for f in os.listdir(scene) if f.endswith('.jpg') or f.endswith('.png'))
            for f in files:
                d = os.path.join(scene, 'depth', f[:-4] + ('.png' if dataset == 'nyu' else '.npy')) if dataset == 'nyu' \
                    else os.path.join(scene, f[:-4] + '.npy')
                if os.path.exists(d):
                    self.imgs.append(os.path.join(scene, f))
                    self.depth.append(d)

    def __getitem__(self, index):
        img = load_as_float(self.imgs[index])
        if self.depth[index].endswith('.npy'):
            depth = torch.from_numpy(np.load(self.depth[index]).astype(np.float32))
        else:
            from PIL import Image
            depth = torch.from_numpy(np.asarray(Image.open(self.depth[index])).astype(np.float32) / 1000)
        if self.transform is not None:
            img, _ = self.transform([img], None)
            img = img[0]
        return img, depth

    def __len__(self):
        return len(self.imgs)
