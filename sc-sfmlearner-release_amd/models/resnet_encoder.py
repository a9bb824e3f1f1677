"""ResNet-18 / -50 encoder with the parameter names of torchvision's ``ResNet`` (conv1, bn1,
layer1..4.<i>.{conv,bn}<k>, downsample.{0,1}, fc), which is what the reference's checkpoints hold under
``encoder.encoder.*`` (models/resnet_encoder.py:17-61; utils.py:57-66).  ``fc`` and ``avgpool`` are
kept although the encoder never calls them -- they are in the state dict."""
from __future__ import absolute_import, division, print_function

import numpy as np
import torch
import torch.nn as nn


def _conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv3x3(planes, planes, stride)  # stride on the 3x3 (torchvision "v1.5")
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


class ResNet(nn.Module):
    """The trunk; ``num_input_images`` frames are stacked on the channel axis of conv1
    (models/resnet_encoder.py:21-23)."""

    def __init__(self, block, layers, num_classes=1000, num_input_images=1):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(num_input_images * 3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


_CFG = {18: (BasicBlock, [2, 2, 2, 2]), 50: (Bottleneck, [3, 4, 6, 3])}


def imagenet_weights_path(num_layers):
    """Local stand-in for the reference's model-zoo download: $SCSFM_IMAGENET_WEIGHTS/resnet<N>.pth or None."""
    import os
    root = os.environ.get("SCSFM_IMAGENET_WEIGHTS")
    if not root:
        return None
    path = os.path.join(root, "resnet{}.pth".format(num_layers))
    return path if os.path.exists(path) else None


class ResnetEncoder(nn.Module):
    """Five feature maps at strides 2..32 (models/resnet_encoder.py:64-97)."""

    def __init__(self, num_layers, pretrained, num_input_images=1):
        super().__init__()
        if num_layers not in _CFG:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        block, layers = _CFG[num_layers]
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4
        self.encoder = ResNet(block, layers, num_input_images=num_input_images)
        if pretrained:
            # the reference downloads ImageNet weights (models/resnet_encoder.py:53-59); offline they come from a
            # local torchvision state dict: $SCSFM_IMAGENET_WEIGHTS/resnet<N>.pth (train.py checks this at
            # argument-parsing time, before any dataset is touched)
            path = imagenet_weights_path(num_layers)
            if path is None:
                raise RuntimeError("ImageNet-pretrained weights are not reachable offline: set SCSFM_IMAGENET_WEIGHTS "
                                   "to a directory holding resnet{}.pth, or use --with-pretrain 0".format(num_layers))
            loaded = torch.load(path, map_location="cpu")
            # several stacked input frames: the first convolution's filters repeated and rescaled (:55-57)
            loaded['conv1.weight'] = torch.cat([loaded['conv1.weight']] * num_input_images, 1) / num_input_images
            self.encoder.load_state_dict(loaded)
        # never reached by forward(): frozen so that DistributedDataParallel does not wait for them
        for p in self.encoder.fc.parameters():
            p.requires_grad_(False)

    def forward(self, input_image):
        e = self.encoder
        f0 = e.relu(e.bn1(e.conv1(input_image)))
        f1 = e.layer1(e.maxpool(f0))
        f2 = e.layer2(f1)
        f3 = e.layer3(f2)
        f4 = e.layer4(f3)
        self.features = [f0, f1, f2, f3, f4]
        return self.features
