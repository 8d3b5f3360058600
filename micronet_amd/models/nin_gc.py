"""Network-in-Network with grouped convs + channel shuffle (CIFAR-10).

Layer table follows the reference ``micronet/models/nin_gc.py:62-147`` (cfg 66);
``ConvBNReLU`` keeps the attribute names ``conv/bn/relu`` the quantisation
rewrite relies on (``models/nin_gc.py:18-59``).
"""
import torch.nn as nn

DEFAULT_CFG = [256, 256, 256, 512, 512, 512, 1024, 1024]

# (cin_idx, cout_idx, kernel, pad, groups, shuffle_groups) ; idx -1 = image (3), idx 8 = classes (10)
#  "P" = 2x2 max-pool
_PLAN = [
    (-1, 0, 5, 2, 1, 0),
    (0, 1, 1, 0, 2, 0),
    (1, 2, 1, 0, 2, 2),
    "P",
    (2, 3, 3, 1, 16, 2),
    (3, 4, 1, 0, 4, 16),
    (4, 5, 1, 0, 4, 4),
    "P",
    (5, 6, 3, 1, 32, 4),
    (6, 7, 1, 0, 8, 32),
    (7, 8, 1, 0, 1, 0),
]


def channel_shuffle(x, groups):
    """(N, g*c, H, W) -> interleave the g groups (ShuffleNet shuffle)."""
    n, ch, h, w = x.size()
    assert ch % groups == 0
    return x.view(n, groups, ch // groups, h, w).transpose(1, 2).contiguous().view(n, ch, h, w)


class ConvBNReLU(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 padding_mode="zeros", eps=1e-5, momentum=0.1, channel_shuffle=0, shuffle_groups=1):
        super().__init__()
        self.channel_shuffle_flag = channel_shuffle
        self.shuffle_groups = shuffle_groups
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                              dilation=dilation, groups=groups, bias=bias, padding_mode=padding_mode)
        self.bn = nn.BatchNorm2d(out_channels, eps=eps, momentum=momentum)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        if self.channel_shuffle_flag:
            x = channel_shuffle(x, groups=self.shuffle_groups)
        return self.relu(self.bn(self.conv(x)))


class Net(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        widths = list(DEFAULT_CFG if cfg is None else cfg)
        ch = lambda i: 3 if i < 0 else (10 if i == 8 else widths[i])
        layers = []
        for item in _PLAN:
            if item == "P":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2, padding=0))
                continue
            ci, co, k, p, g, sg = item
            layers.append(ConvBNReLU(ch(ci), ch(co), kernel_size=k, stride=1, padding=p, groups=g,
                                     channel_shuffle=1 if sg else 0, shuffle_groups=sg if sg else 1))
        layers.append(nn.AvgPool2d(kernel_size=8, stride=1, padding=0))
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        x = self.model(x)
        return x.view(x.size(0), -1)
