"""Drop-in import surface: the module paths of 666DZY666/micronet's quantization hot path, served by ``micronet_amd``.

``micronet.compression.quantization.{wbwtab,wqaq.dorefa,wqaq.iao}.quantize``, ``micronet.base_module.op`` and
``micronet.models.{nin,nin_gc,resnet}`` resolve to the MI355X-native implementation, so code written against the
reference (its ``main.py`` training scripts, the README transfer demo) imports unchanged.  See INTEGRATION.md.
"""
__version__ = "1.12.0+mi355x"

from micronet.base_module.op import *  # noqa: F401,F403


def _lenet():
    import torch.nn as nn
    import torch.nn.functional as F

    class LeNet(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(1, 10, kernel_size=5)
            self.conv2 = nn.Conv2d(10, 20, kernel_size=5)
            self.fc1 = nn.Linear(320, 50)
            self.fc2 = nn.Linear(50, 10)
            self.max_pool = nn.MaxPool2d(kernel_size=2)
            self.relu = nn.ReLU(inplace=True)

        def forward(self, x):
            x = self.relu(self.max_pool(self.conv1(x)))
            x = self.relu(self.max_pool(self.conv2(x)))
            x = self.relu(self.fc1(x.view(-1, 320)))
            return F.log_softmax(self.fc2(F.dropout(x, training=self.training)), dim=1)

    return LeNet()


def quant_test_auto():
    """Import-surface smoke test (counterpart of the reference's micronet/__init__.py:128-175): rewrite a LeNet with
    each scheme's ``prepare`` and print the result.  Construction only -- no kernels run."""
    import micronet.compression.quantization.wbwtab.quantize as quant_wbwtab
    import micronet.compression.quantization.wqaq.dorefa.quantize as quant_dorefa
    import micronet.compression.quantization.wqaq.iao.quantize as quant_iao

    lenet = _lenet()
    out = {
        "wbwtab": quant_wbwtab.prepare(lenet, inplace=False),
        "dorefa": quant_dorefa.prepare(lenet, inplace=False),
        "iao": quant_iao.prepare(lenet, inplace=False),
    }
    for name, m in out.items():
        print("***quant_lenet_%s***\n" % name, m)
    print("\nquant_model is ready")
    print("micronet is ready")
    return out


def quant_test_manual():
    """Counterpart of micronet/__init__.py:6-125: build the quantised LeNets by hand from the quant ops."""
    import torch.nn as nn
    from micronet.compression.quantization.wbwtab.quantize import ActivationQuantizer as relu_wbwtab
    from micronet.compression.quantization.wbwtab.quantize import QuantConv2d as conv_wbwtab
    from micronet.compression.quantization.wqaq.dorefa.quantize import QuantConv2d as conv_dorefa
    from micronet.compression.quantization.wqaq.dorefa.quantize import QuantLinear as linear_dorefa
    from micronet.compression.quantization.wqaq.iao.quantize import QuantConv2d as conv_iao
    from micronet.compression.quantization.wqaq.iao.quantize import QuantLinear as linear_iao
    from micronet.compression.quantization.wqaq.iao.quantize import QuantMaxPool2d as max_pool_iao
    from micronet.compression.quantization.wqaq.iao.quantize import QuantReLU as relu_iao

    def build(conv, linear, pool, relu):
        return nn.ModuleDict(dict(conv1=conv(1, 10, kernel_size=5), conv2=conv(10, 20, kernel_size=5),
                                  fc1=linear(320, 50), fc2=linear(50, 10), max_pool=pool(kernel_size=2), relu=relu()))

    out = {
        "wbwtab": build(conv_wbwtab, nn.Linear, nn.MaxPool2d, relu_wbwtab),
        "dorefa": build(conv_dorefa, linear_dorefa, nn.MaxPool2d, lambda: nn.ReLU(inplace=True)),
        "iao": build(conv_iao, linear_iao, max_pool_iao, lambda: relu_iao(inplace=True)),
    }
    for name, m in out.items():
        print("***quant_lenet_%s***\n" % name, m)
    print("\nquant_model is ready")
    print("micronet is ready")
    return out
