"""Checkpoint writer and small helpers with the reference's names (utils.py).  The checkpoint format
is the reference's (utils.py:57-66, train.py:219-227): ``<prefix>_checkpoint.pth.tar`` =
``torch.save({'epoch': e + 1, 'state_dict': net.state_dict()})`` for prefix in {dispnet, exp_pose},
copied to ``<prefix>_model_best.pth.tar`` when the decisive error improves.  No optimiser state."""
from __future__ import division

import os
import shutil

import numpy as np
import torch


def save_checkpoint(save_path, dispnet_state, exp_pose_state, is_best, filename='checkpoint.pth.tar'):
    file_prefixes = ['dispnet', 'exp_pose']
    states = [dispnet_state, exp_pose_state]
    for prefix, state in zip(file_prefixes, states):
        torch.save(state, os.path.join(str(save_path), '{}_{}'.format(prefix, filename)))
    if is_best:
        for prefix in file_prefixes:
            shutil.copyfile(os.path.join(str(save_path), '{}_{}'.format(prefix, filename)),
                            os.path.join(str(save_path), '{}_model_best.pth.tar'.format(prefix)))


def tensor2array(tensor, max_value=None, colormap='rainbow'):
    """[H,W] / [1,H,W] map -> 3xHxW pseudo-colour array in [0,1]; [3,H,W] image -> de-normalised
    array (utils.py:42-54).  matplotlib is optional here: without it a grey ramp is used."""
    tensor = tensor.detach().cpu()
    if max_value is None:
        max_value = tensor.max().item()
    if tensor.ndimension() == 2 or tensor.size(0) == 1:
        norm = (tensor.squeeze().numpy() / max(max_value, 1e-12)).clip(0, 1)
        try:
            from matplotlib import cm
            cmap = {'rainbow': cm.rainbow, 'magma': getattr(cm, 'magma', cm.rainbow), 'bone': cm.bone}.get(colormap, cm.rainbow)
            array = cmap(norm).astype(np.float32)[:, :, :3].transpose(2, 0, 1)
        except Exception:
            array = np.stack([norm] * 3, 0).astype(np.float32)
    else:
        assert tensor.size(0) == 3
        array = 0.45 + tensor.numpy() * 0.225
    return array
