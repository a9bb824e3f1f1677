"""Synthetic stand-ins for KITTI / NYU (no dataset is reachable offline).

``write_sequence_tree`` writes a small tree in the reference's SequenceFolder layout (scene/NNNNNNN.jpg,
cam.txt, optional NNNNNNN.npy depth, train.txt, val.txt) so that train.py's data path can be exercised
end to end; ``InMemorySequences`` serves normalised tensors straight from memory for throughput runs
where JPEG decoding on the host would be the bottleneck."""
import os

import numpy as np
import torch
import torch.utils.data as data
from PIL import Image

KITTI_K = np.array([[721.5377, 0, 609.5593], [0, 721.5377, 172.854], [0, 0, 1]], dtype=np.float32)


def _frame(rng, h, w, shift):
    """A smooth random scene panned by `shift` pixels, plus texture."""
    base = rng.random((h // 8 + 2, (w + 64) // 8 + 2, 3)).astype(np.float32)
    img = np.asarray(Image.fromarray((base * 255).astype(np.uint8)).resize((w + 64, h), Image.BILINEAR)).astype(np.float32)
    img = img[:, shift:shift + w] * 0.9 + rng.random((h, w, 3)).astype(np.float32) * 25
    return img.clip(0, 255).astype(np.uint8)


def write_sequence_tree(root, n_scenes=2, frames_per_scene=8, height=256, width=832, seed=0, with_depth=True):
    rng = np.random.default_rng(seed)
    os.makedirs(root, exist_ok=True)
    names = []
    K = KITTI_K.copy()
    K[0] *= width / 1242.0
    K[1] *= height / 375.0
    for s in range(n_scenes):
        name = 'scene_{:02d}'.format(s)
        names.append(name)
        d = os.path.join(root, name)
        os.makedirs(d, exist_ok=True)
        np.savetxt(os.path.join(d, 'cam.txt'), K)
        scene_rng = np.random.default_rng(seed * 1000 + s)
        state = scene_rng.bit_generator.state
        for f in range(frames_per_scene):
            scene_rng.bit_generator.state = state  # same scene, panned
            Image.fromarray(_frame(scene_rng, height, width, 2 * f)).save(os.path.join(d, '{:07d}.jpg'.format(f)), quality=92)
            if with_depth:
                depth = 1.0 / (10 * rng.random((height // 16 + 1, width // 16 + 1)).astype(np.float32) + 0.02)
                depth = np.asarray(Image.fromarray(depth).resize((width, height), Image.BILINEAR))
                np.save(os.path.join(d, '{:07d}.npy'.format(f)), depth.astype(np.float32))
    with open(os.path.join(root, 'train.txt'), 'w') as f:
        f.write(''.join(n + '\n' for n in names[:max(1, n_scenes - 1)]))
    with open(os.path.join(root, 'val.txt'), 'w') as f:
        f.write(names[-1] + '\n')
    return root


class InMemorySequences(data.Dataset):
    """(tgt_img, [ref_imgs], K, K^-1) samples of normalised random tensors."""

    def __init__(self, n_samples, height=256, width=832, sequence_length=3, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.imgs = (torch.rand(n_samples, sequence_length, 3, height, width, generator=g) - 0.45) / 0.225
        K = KITTI_K.copy()
        K[0] *= width / 1242.0
        K[1] *= height / 375.0
        self.K = K
        self.scenes = ['synthetic']

    def __getitem__(self, i):
        fr = self.imgs[i]
        mid = fr.shape[0] // 2
        refs = [fr[j] for j in range(fr.shape[0]) if j != mid]
        return fr[mid], refs, self.K, np.linalg.inv(self.K)

    def __len__(self):
        return self.imgs.shape[0]
