"""CIFAR-style ResNets whose residual add is an ``Add`` module (so iao can quantise it).

Module tree / names follow the reference ``micronet/models/resnet.py``
(BasicBlock 7-65, BottleNeck 68-119, ResNet 122-177, factories 180-202):
``conv1`` stem without max-pool, stages ``conv2_x..conv5_x``, ``avg_pool``, ``fc``.
"""
import torch.nn as nn

from micronet_amd.base_module.op import Add


def _cbr(cin, cout, k, stride=1, relu=True):
    mods = [nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False), nn.BatchNorm2d(cout)]
    if relu:
        mods.append(nn.ReLU(inplace=True))
    return mods


class _Residual(nn.Module):
    expansion = 1

    def _finish(self, in_channels, out_channels, stride):
        outc = out_channels * self.expansion
        self.shortcut = nn.Sequential()
        if stride != 1 or in_channels != outc:
            self.shortcut = nn.Sequential(
                nn.Conv2d(in_channels, outc, kernel_size=1, stride=stride, bias=False), nn.BatchNorm2d(outc))
        self.add = Add()

    def forward(self, x):
        # the trailing ReLU is created inline (not a child), exactly as the reference does (resnet.py:63)
        return nn.ReLU(inplace=True)(self.add(self.residual_function(x), self.shortcut(x)))


class BasicBlock(_Residual):
    expansion = 1

    def __init__(self, in_channels, out_channels, stride=1):
        super().__init__()
        self.residual_function = nn.Sequential(
            *_cbr(in_channels, out_channels, 3, stride), *_cbr(out_channels, out_channels * self.expansion, 3, relu=False))
        self._finish(in_channels, out_channels, stride)


class BottleNeck(_Residual):
    expansion = 4

    def __init__(self, in_channels, out_channels, stride=1):
        super().__init__()
        self.residual_function = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=False), nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, stride=stride, kernel_size=3, padding=1, bias=False),
            nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels * self.expansion, kernel_size=1, bias=False),
            nn.BatchNorm2d(out_channels * self.expansion))
        self._finish(in_channels, out_channels, stride)


class ResNet(nn.Module):
    def __init__(self, block, num_block, num_classes=10):
        super().__init__()
        self.in_channels = 64
        self.conv1 = nn.Sequential(*_cbr(3, 64, 3))
        for i, (width, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2))):
            setattr(self, "conv%d_x" % (i + 2), self._make_layer(block, width, num_block[i], stride))
        self.avg_pool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)

    def _make_layer(self, block, out_channels, num_blocks, stride):
        blocks = []
        for s in [stride] + [1] * (num_blocks - 1):
            blocks.append(block(self.in_channels, out_channels, s))
            self.in_channels = out_channels * block.expansion
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = self.conv1(x)
        for i in range(2, 6):
            x = getattr(self, "conv%d_x" % i)(x)
        x = self.avg_pool(x)
        return self.fc(x.view(x.size(0), -1))


def resnet18():
    return ResNet(BasicBlock, [2, 2, 2, 2])


def resnet34():
    return ResNet(BasicBlock, [3, 4, 6, 3])


def resnet50():
    return ResNet(BottleNeck, [3, 4, 6, 3])


def resnet101():
    return ResNet(BottleNeck, [3, 4, 23, 3])


def resnet152():
    return ResNet(BottleNeck, [3, 8, 36, 3])
