"""SequenceFolder with the reference's on-disk layout and sample structure
(datasets/sequence_folders.py): root/{train,val}.txt list scene folders holding NNNNNNN.jpg frames and
a cam.txt (3x3 intrinsics).  A sample is (tgt_img, [ref_imgs], K, K^-1).  Frames are decoded with PIL
(the reference uses imageio, absent here).  `datasets.synthetic` generates such a tree."""
import os
import random

import numpy as np
import torch.utils.data as data
from PIL import Image


def load_as_float(path):
    return np.asarray(Image.open(path).convert('RGB')).astype(np.float32)


class SequenceFolder(data.Dataset):
    def __init__(self, root, seed=None, train=True, sequence_length=3, transform=None, skip_frames=1, dataset='kitti'):
        np.random.seed(seed)
        random.seed(seed)
        self.root = str(root)
        scene_list = os.path.join(self.root, 'train.txt' if train else 'val.txt')
        self.scenes = [os.path.join(self.root, line.strip()) for line in open(scene_list) if line.strip()]
        self.transform = transform
        self.dataset = dataset
        self.k = skip_frames
        self.crawl_folders(sequence_length)

    def crawl_folders(self, sequence_length):
        samples = []
        demi = (sequence_length - 1) // 2
        shifts = list(range(-demi * self.k, demi * self.k + 1, self.k))
        shifts.pop(demi)
        for scene in self.scenes:
            intrinsics = np.genfromtxt(os.path.join(scene, 'cam.txt')).astype(np.float32).reshape((3, 3))
            imgs = sorted(os.path.join(scene, f) for f in os.listdir(scene) if f.endswith('.jpg') or f.endswith('.png'))
            if len(imgs) < sequence_length:
                continue
            for i in range(demi * self.k, len(imgs) - demi * self.k):
                samples.append({'intrinsics': intrinsics, 'tgt': imgs[i], 'ref_imgs': [imgs[i + j] for j in shifts]})
        random.shuffle(samples)
        self.samples = samples

    def __getitem__(self, index):
        sample = self.samples[index]
        tgt_img = load_as_float(sample['tgt'])
        ref_imgs = [load_as_float(r) for r in sample['ref_imgs']]
        if self.transform is not None:
            imgs, intrinsics = self.transform([tgt_img] + ref_imgs, np.copy(sample['intrinsics']))
            tgt_img, ref_imgs = imgs[0], imgs[1:]
        else:
            intrinsics = np.copy(sample['intrinsics'])
        return tgt_img, ref_imgs, intrinsics, np.linalg.inv(intrinsics)

    def __len__(self):
        return len(self.samples)
