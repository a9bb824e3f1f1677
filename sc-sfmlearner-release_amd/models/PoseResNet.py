"""PoseResNet: two stacked frames -> 6-DoF pose, 0.01 * mean over the feature map
(models/PoseResNet.py:14-66).  State-dict layout: ``encoder.encoder.*`` (conv1 takes 6 channels) and
``decoder.net.{0: squeeze, 1..3: pose convs}``."""
from __future__ import absolute_import, division, print_function

import torch
import torch.nn as nn

from .resnet_encoder import ResnetEncoder


class PoseDecoder(nn.Module):
    def __init__(self, num_ch_enc, num_input_features=1, num_frames_to_predict_for=1, stride=1):
        super().__init__()
        self.net = nn.ModuleList([
            nn.Conv2d(int(num_ch_enc[-1]), 256, 1),
            nn.Conv2d(num_input_features * 256, 256, 3, stride, 1),
            nn.Conv2d(256, 256, 3, stride, 1),
            nn.Conv2d(256, 6 * num_frames_to_predict_for, 1)])
        self.relu = nn.ReLU()

    def forward(self, input_features):
        out = torch.cat([self.relu(self.net[0](f[-1])) for f in input_features], 1)
        out = self.relu(self.net[1](out))
        out = self.relu(self.net[2](out))
        out = self.net[3](out)
        return 0.01 * out.mean(3).mean(2).view(-1, 6)


class PoseResNet(nn.Module):
    def __init__(self, num_layers=18, pretrained=True):
        super().__init__()
        self.encoder = ResnetEncoder(num_layers=num_layers, pretrained=pretrained, num_input_images=2)
        self.decoder = PoseDecoder(self.encoder.num_ch_enc)

    def init_weights(self):
        pass

    def forward(self, img1, img2):
        return self.decoder([self.encoder(torch.cat([img1, img2], 1))])
