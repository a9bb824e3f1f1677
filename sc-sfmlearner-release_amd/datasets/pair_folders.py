"""PairFolder (datasets/pair_folders.py): the behaviour of the reference's loader for pair data sets.

On-disk contract (datasets/pair_folders.py:14-21,33-45 of the reference): a scene folder holds the frames as
``*.jpg`` and ONE intrinsics file ``*.txt`` PER PAIR; both lists are taken in sorted order, frames 2k and 2k+1 form
pair k (target = the first, reference = the second, never swapped) and pair k uses the k-th text file.  A trailing
unpaired frame is ignored.  The pairs of all scenes are shuffled once with the seeded generator.
"""
import os
import random

import numpy as np
import torch.utils.data as data

from .sequence_folders import load_as_float


def _sorted_with_suffix(folder, suffix):
    return sorted(os.path.join(folder, f) for f in os.listdir(folder) if f.endswith(suffix))


class PairFolder(data.Dataset):
    def __init__(self, root, seed=None, train=True, transform=None):
        np.random.seed(seed)
        random.seed(seed)
        self.root = str(root)
        listing = os.path.join(self.root, 'train.txt' if train else 'val.txt')
        self.scenes = [os.path.join(self.root, line[:-1] if line.endswith('\n') else line) for line in open(listing)]
        self.transform = transform
        self.samples = self._collect()

    def _collect(self):
        found = []
        for scene in self.scenes:
            frames = _sorted_with_suffix(scene, '.jpg')
            cams = _sorted_with_suffix(scene, '.txt')
            for k in range(len(frames) // 2):
                K = np.genfromtxt(cams[k]).astype(np.float32).reshape((3, 3))
                found.append({'intrinsics': K, 'tgt': frames[2 * k], 'ref_imgs': [frames[2 * k + 1]]})
        random.shuffle(found)
        return found

    def __getitem__(self, index):
        entry = self.samples[index]
        frames = [load_as_float(entry['tgt'])] + [load_as_float(p) for p in entry['ref_imgs']]
        K = np.copy(entry['intrinsics'])
        if self.transform is not None:
            frames, K = self.transform(frames, K)
        return frames[0], frames[1:], K, np.linalg.inv(K)

    def __len__(self):
        return len(self.samples)
