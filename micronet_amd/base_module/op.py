"""Residual-add as a module so ``prepare()`` can find and quantise it.

Mirror of the reference's ``micronet/base_module/op.py:5-12`` (``Add``): the iao
graph rewrite replaces every ``Add`` child with ``QuantAdd``.
"""
import torch.nn as nn

__all__ = ["Add"]


class Add(nn.Module):
    def forward(self, res, shortcut):
        return res + shortcut
