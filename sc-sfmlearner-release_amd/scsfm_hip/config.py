"""Run-time knobs of the HIP loss path."""
from __future__ import annotations

# (w_photo, w_geom): the weights the training loop multiplies photo_loss and geometry_loss with
# (train.py:268, defaults -p 1 -c 0.5 of scripts/train_resnet18_depth_256.sh).  With a hint the
# forward of compute_photo_and_geometry_loss already runs the backward's tiled pass (speculative
# forward, see include/scsfm_hip.h: scsfm_pair_fwd_spec).  A wrong hint costs time, never
# correctness: the backward checks the actual upstream gradients on the device and recomputes --
# and leaves them in the device-side copy of the hint (scsfm_pair_desc::hint), so the step after
# a mis-speculated one already speculates on the weights the loop really uses.  set_weight_hint
# is therefore an optimisation of the FIRST step only.
_hint = (1.0, 0.5)
_hint_dev = {}  # device -> float64[2] tensor holding (w_photo, w_geom); rewritten by every backward


def set_weight_hint(w_photo, w_geom):
    """Tell the loss which upstream-gradient ratio to speculate on; ``None, None`` disables it."""
    global _hint
    if w_photo is None:
        # (the device tensors stay alive: a captured HIP graph may still hold their addresses; hint_tensor() hands
        # none out while speculation is off)
        _hint = None
        return
    # the upstream gradients arrive as fp32 tensors: compare against the fp32 roundings of the weights
    # (the device-side check is an exact equality of products)
    import numpy as np
    _hint = (float(np.float32(w_photo)), float(np.float32(w_geom)))
    for t in _hint_dev.values():
        t.copy_(t.new_tensor(_hint))


def weight_hint():
    return _hint


def hint_tensor(device):
    """The device-side (w_photo, w_geom) of ``device`` (created from the host hint on first use); None when speculation
    is disabled."""
    if _hint is None:
        return None
    import torch
    key = (device.type, device.index)
    t = _hint_dev.get(key)
    if t is None:
        t = torch.tensor(_hint, dtype=torch.float64, device=device)
        _hint_dev[key] = t
    return t


def check_window():
    """SCSFM_CHECK_WINDOW=1 (debugging): every speculative forward counts wraps of its fixed-point scatter cells and the
    call raises capi.WindowOverflow if there are any (synchronises; the runtime-flag kernel instantiation).  Read per
    call."""
    import os
    return os.environ.get("SCSFM_CHECK_WINDOW") == "1"


_smooth_rides = None


def smooth_rides_along():
    """Does the speculative forward of compute_photo_and_geometry_loss also evaluate the smooth loss of its frames (round 6:
    scsfm_pair_desc::smooth_ws; the compute_smooth_loss call that follows then finds its result waiting)?  Default on;
    SCSFM_SMOOTH_RIDE=0 (read once) or set_smooth_rides_along(False) restore the stand-alone smooth forward."""
    global _smooth_rides
    if _smooth_rides is None:
        import os
        _smooth_rides = os.environ.get("SCSFM_SMOOTH_RIDE", "1") != "0"
    return _smooth_rides


def set_smooth_rides_along(on):
    global _smooth_rides
    _smooth_rides = bool(on)
