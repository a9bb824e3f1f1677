"""DispResNet / PoseResNet in plain PyTorch (the reference builds them on torchvision, which this
image does not ship).  Module names and parameter shapes reproduce the reference's state dicts
(models/DispResNet.py, models/PoseResNet.py, models/resnet_encoder.py), so checkpoints are
interchangeable; the convolutions run on PyTorch-ROCm / MIOpen (MFMA), as north_star prescribes."""
from .DispResNet import DispResNet
from .PoseResNet import PoseResNet

__all__ = ["DispResNet", "PoseResNet"]
