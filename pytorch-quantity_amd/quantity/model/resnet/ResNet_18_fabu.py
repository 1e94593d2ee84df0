"""CIFAR-10 ResNet-18 in "fabu" style: residual adds and the flatten are marker MODULES (Eltwise, View)
and every ReLU is out-of-place, so forward hooks see each cared tensor.  Fixture model of BASELINE
config 1.

The module tree (attribute names, Sequential indices, registration order) equals the reference's
quantity/model/resnet/ResNet_18_fabu.py (ResidualBlock :11-36, ResNet :38-70, ResNet18 :72), because
feat.table / weight.table rows are keyed by named_modules() names: conv1.0, layer1.0.left.0,
layer1.0.left.3, layer1.0.Eltwise, layer2.0.shortcut.0, ..., fc.
"""
import sys

import torch.nn as nn

sys.path.insert(0, '../../')
from common.quantity import Eltwise, View  # noqa: E402

STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))        # (channels, stride of the first block)
BLOCKS_PER_STAGE = 2


def _conv_bn(cin, cout, k, stride, pad):
    return [nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad, bias=False), nn.BatchNorm2d(cout)]


class ResidualBlock(nn.Module):
    """relu(Eltwise(left(x), shortcut(x))); left = conv3x3-bn-relu-conv3x3-bn, shortcut = identity or
    a strided 1x1 conv-bn when the shape changes."""

    def __init__(self, inchannel, outchannel, stride=1):
        super(ResidualBlock, self).__init__()
        trunk = _conv_bn(inchannel, outchannel, 3, stride, 1)
        trunk.append(nn.ReLU(False))
        trunk.extend(_conv_bn(outchannel, outchannel, 3, 1, 1))
        needs_projection = (stride != 1) or (inchannel != outchannel)
        self.left = nn.Sequential(*trunk)
        self.shortcut = nn.Sequential(*(_conv_bn(inchannel, outchannel, 1, stride, 0) if needs_projection else ()))
        self.Eltwise = Eltwise()
        self.relu = nn.ReLU(False)

    def forward(self, x):
        summed = self.Eltwise(self.left(x), self.shortcut(x))
        return self.relu(summed)


class ResNet(nn.Module):

    def __init__(self, ResidualBlock, num_classes=10):
        super(ResNet, self).__init__()
        self.inchannel = STAGES[0][0]
        self.conv1 = nn.Sequential(*_conv_bn(3, self.inchannel, 3, 1, 1), nn.ReLU(False))
        for idx, (width, stride) in enumerate(STAGES, start=1):
            setattr(self, "layer%d" % idx, self.make_layer(ResidualBlock, width, BLOCKS_PER_STAGE, stride))
        self.avePool2d = nn.AvgPool2d(4)
        self.fc = nn.Linear(STAGES[-1][0], num_classes)
        self.view = View()

    def make_layer(self, block, channels, num_blocks, stride):
        stage = nn.Sequential()
        for n in range(num_blocks):
            stage.add_module(str(n), block(self.inchannel, channels, stride if n == 0 else 1))
            self.inchannel = channels
        return stage

    def forward(self, x):
        feat = self.conv1(x)
        for idx in range(1, len(STAGES) + 1):
            feat = getattr(self, "layer%d" % idx)(feat)
        return self.fc(self.view(self.avePool2d(feat)))


def ResNet18():
    return ResNet(ResidualBlock)
