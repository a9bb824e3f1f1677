"""DispResNet: ResNet encoder + monodepth2-style decoder, disparity = 10 * sigmoid + 0.01 at four
scales (models/DispResNet.py:49-121).  State-dict layout: ``encoder.encoder.*`` and
``decoder.decoder.<k>`` with k = 0..9 the (upconv i, j) blocks for i = 4..0, j = 0..1 and k = 10..13
the disparity heads of scales 0..3."""
from __future__ import absolute_import, division, print_function

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .resnet_encoder import ResnetEncoder


class Conv3x3(nn.Module):
    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.pad = nn.ReflectionPad2d(1) if use_refl else nn.ZeroPad2d(1)
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)

    def forward(self, x):
        return self.conv(self.pad(x))


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)
        self.nonlin = nn.ELU(inplace=True)

    def forward(self, x):
        return self.nonlin(self.conv(x))


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.alpha, self.beta = 10, 0.01
        self.use_skips = use_skips
        self.scales = list(scales)
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        blocks, self._up, self._head = [], {}, {}
        for i in range(4, -1, -1):
            cin = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self._up[(i, 0)] = len(blocks)
            blocks.append(ConvBlock(cin, self.num_ch_dec[i]))
            cin = self.num_ch_dec[i] + (self.num_ch_enc[i - 1] if use_skips and i > 0 else 0)
            self._up[(i, 1)] = len(blocks)
            blocks.append(ConvBlock(cin, self.num_ch_dec[i]))
        for s in self.scales:
            self._head[s] = len(blocks)
            blocks.append(Conv3x3(self.num_ch_dec[s], num_output_channels))
        self.decoder = nn.ModuleList(blocks)
        self.sigmoid = nn.Sigmoid()

    def forward(self, feats):
        outputs = []
        x = feats[-1]
        for i in range(4, -1, -1):
            x = self.decoder[self._up[(i, 0)]](x)
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            if self.use_skips and i > 0:
                x = torch.cat([x, feats[i - 1]], 1)
            x = self.decoder[self._up[(i, 1)]](x)
            if i in self._head:
                outputs.append(self.alpha * self.sigmoid(self.decoder[self._head[i]](x)) + self.beta)
        return outputs[::-1]


class DispResNet(nn.Module):
    def __init__(self, num_layers=18, pretrained=True):
        super().__init__()
        self.encoder = ResnetEncoder(num_layers=num_layers, pretrained=pretrained, num_input_images=1)
        self.decoder = DepthDecoder(self.encoder.num_ch_enc)

    def init_weights(self):
        pass

    def forward(self, x):
        outputs = self.decoder(self.encoder(x))
        return outputs if self.training else outputs[0]
