"""CIFAR-10 ResNet-18 written with marker layers (Eltwise / View modules, non-inplace ReLU) so that
every add / flatten is a hookable nn.Module -- the fixture model of BASELINE config 1.

Same module tree as the reference's quantity/model/resnet/ResNet_18_fabu.py (ResidualBlock :11-36,
ResNet :38-70, ResNet18 :72): attribute names and registration order are part of the drop-in
surface because feat.table / weight.table rows are keyed by named_modules() names
(conv1.0, layer1.0.left.0, layer1.0.Eltwise, ..., fc).
"""
import sys

import torch.nn as nn

sys.path.insert(0, '../../')
from common.quantity import Eltwise, View  # noqa: E402


def _conv_bn(cin, cout, k, stride, pad):
    return [nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=False), nn.BatchNorm2d(cout)]


class ResidualBlock(nn.Module):
    """left: 3x3 conv-bn-relu-3x3 conv-bn; shortcut: identity or 1x1 conv-bn; Eltwise; relu."""

    def __init__(self, inchannel, outchannel, stride=1):
        super(ResidualBlock, self).__init__()
        body = _conv_bn(inchannel, outchannel, 3, stride, 1) + [nn.ReLU(False)] + _conv_bn(outchannel, outchannel, 3, 1, 1)
        self.left = nn.Sequential(*body)
        projected = stride != 1 or inchannel != outchannel
        self.shortcut = nn.Sequential(*(_conv_bn(inchannel, outchannel, 1, stride, 0) if projected else []))
        self.Eltwise = Eltwise()
        self.relu = nn.ReLU(False)

    def forward(self, x):
        return self.relu(self.Eltwise(self.left(x), self.shortcut(x)))


class ResNet(nn.Module):

    def __init__(self, ResidualBlock, num_classes=10):
        super(ResNet, self).__init__()
        self.inchannel = 64
        self.conv1 = nn.Sequential(*(_conv_bn(3, 64, 3, 1, 1) + [nn.ReLU(False)]))
        self.layer1 = self.make_layer(ResidualBlock, 64, 2, stride=1)
        self.layer2 = self.make_layer(ResidualBlock, 128, 2, stride=2)
        self.layer3 = self.make_layer(ResidualBlock, 256, 2, stride=2)
        self.layer4 = self.make_layer(ResidualBlock, 512, 2, stride=2)
        self.avePool2d = nn.AvgPool2d(4)
        self.fc = nn.Linear(512, num_classes)
        self.view = View()

    def make_layer(self, block, channels, num_blocks, stride):
        blocks = []
        for s in [stride] + [1] * (num_blocks - 1):
            blocks.append(block(self.inchannel, channels, s))
            self.inchannel = channels
        return nn.Sequential(*blocks)

    def forward(self, x):
        out = self.conv1(x)
        for stage in (self.layer1, self.layer2, self.layer3, self.layer4):
            out = stage(out)
        out = self.view(self.avePool2d(out))
        return self.fc(out)


def ResNet18():
    return ResNet(ResidualBlock)
