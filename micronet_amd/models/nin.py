"""Plain Network-in-Network (CIFAR-10); reference ``micronet/models/nin.py:42-65``."""
import torch.nn as nn

DEFAULT_CFG = [192, 160, 96, 192, 192, 192, 192, 192]
# (cin_idx, cout_idx, kernel, pad); "P" = 3x3/s2/p1 max-pool
_PLAN = [(-1, 0, 5, 2), (0, 1, 1, 0), (1, 2, 1, 0), "P", (2, 3, 5, 2), (3, 4, 1, 0), (4, 5, 1, 0), "P",
         (5, 6, 3, 1), (6, 7, 1, 0), (7, 8, 1, 0)]


class ConvBNReLU(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 padding_mode="zeros", eps=1e-5, momentum=0.1):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                              dilation=dilation, groups=groups, bias=bias, padding_mode=padding_mode)
        self.bn = nn.BatchNorm2d(out_channels, eps=eps, momentum=momentum)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.relu(self.bn(self.conv(x)))


class Net(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        widths = list(DEFAULT_CFG if cfg is None else cfg)
        ch = lambda i: 3 if i < 0 else (10 if i == 8 else widths[i])
        layers = []
        for item in _PLAN:
            if item == "P":
                layers.append(nn.MaxPool2d(kernel_size=3, stride=2, padding=1))
            else:
                ci, co, k, p = item
                layers.append(ConvBNReLU(ch(ci), ch(co), kernel_size=k, stride=1, padding=p))
        layers.append(nn.AvgPool2d(kernel_size=8, stride=1, padding=0))
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        x = self.model(x)
        return x.view(x.size(0), -1)
