"""ImageNet-style bottleneck ResNet-50 / ResNet-101 written with marker layers, so that the
calibrator can see every residual add (BASELINE configs 2-5).

The reference ships only a torchvision-style bottleneck net (quantity/model/resnet/ResNet.py:31-61,
:161-217, :242-261) that uses `out += residual` and in-place ReLU and therefore cannot be calibrated
by the tool (reference README.md:45-47).  This file is the same topology rebuilt the "fabu" way:
Eltwise module for the add, one non-inplace ReLU module per use, View + AvgPool2d head, and modules
registered in execution order (the table writer pairs named_modules() order with execution order).

Cared tensors: image + 53 conv + 1 fc + 16 Eltwise = 71 (ResNet-50); 139 for ResNet-101.
"""
import sys

import torch.nn as nn

sys.path.insert(0, '../../')
from common.quantity import Eltwise, View  # noqa: E402


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, project=False):
        super(Bottleneck, self).__init__()
        out_planes = planes * self.expansion
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu1 = nn.ReLU(False)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu2 = nn.ReLU(False)
        self.conv3 = nn.Conv2d(planes, out_planes, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(out_planes)
        if project:
            self.downsample = nn.Sequential(
                nn.Conv2d(inplanes, out_planes, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(out_planes))
        else:
            self.downsample = nn.Sequential()
        self.Eltwise = Eltwise()
        self.relu3 = nn.ReLU(False)

    def forward(self, x):
        y = self.relu1(self.bn1(self.conv1(x)))
        y = self.relu2(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu3(self.Eltwise(y, self.downsample(x)))


class ResNetFabu(nn.Module):

    def __init__(self, layers, num_classes=1000, input_size=224):
        super(ResNetFabu, self).__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(False)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AvgPool2d(max(input_size // 32, 1))
        self.view = View()
        self.fc = nn.Linear(512 * Bottleneck.expansion, num_classes)

    def _make_layer(self, planes, blocks, stride):
        project = stride != 1 or self.inplanes != planes * Bottleneck.expansion
        stage = [Bottleneck(self.inplanes, planes, stride, project)]
        self.inplanes = planes * Bottleneck.expansion
        stage += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*stage)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(self.view(self.avgpool(x)))


def ResNet50(num_classes=1000, input_size=224):
    return ResNetFabu([3, 4, 6, 3], num_classes, input_size)


def ResNet101(num_classes=1000, input_size=224):
    return ResNetFabu([3, 4, 23, 3], num_classes, input_size)
