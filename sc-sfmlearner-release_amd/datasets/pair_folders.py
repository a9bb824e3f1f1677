"""PairFolder (datasets/pair_folders.py): scenes hold image pairs NNNN_0.jpg / NNNN_1.jpg; a sample is
(tgt, [ref], K, K^-1) with the two frames in random order."""
import os
import random

import numpy as np
import torch.utils.data as data

from .sequence_folders import load_as_float


class PairFolder(data.Dataset):
    def __init__(self, root, seed=None, train=True, transform=None):
        np.random.seed(seed)
        random.seed(seed)
        self.root = str(root)
        scene_list = os.path.join(self.root, 'train.txt' if train else 'val.txt')
        self.scenes = [os.path.join(self.root, line.strip()) for line in open(scene_list) if line.strip()]
        self.transform = transform
        pairs = []
        for scene in self.scenes:
            intrinsics = np.genfromtxt(os.path.join(scene, 'cam.txt')).astype(np.float32).reshape((3, 3))
            firsts = sorted(f for f in os.listdir(scene) if f.endswith('_0.jpg'))
            for f in firsts:
                second = os.path.join(scene, f[:-6] + '_1.jpg')
                if os.path.exists(second):
                    pairs.append({'intrinsics': intrinsics, 'a': os.path.join(scene, f), 'b': second})
        random.shuffle(pairs)
        self.samples = pairs

    def __getitem__(self, index):
        s = self.samples[index]
        a, b = load_as_float(s['a']), load_as_float(s['b'])
        if random.random() < 0.5:
            a, b = b, a
        if self.transform is not None:
            imgs, intrinsics = self.transform([a, b], np.copy(s['intrinsics']))
            a, b = imgs
        else:
            intrinsics = np.copy(s['intrinsics'])
        return a, [b], intrinsics, np.linalg.inv(intrinsics)

    def __len__(self):
        return len(self.samples)
