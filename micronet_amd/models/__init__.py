"""CIFAR-10 benchmark nets that define the hot-path layer shapes.

Same module tree / parameter names / construction order as the reference's
``micronet/models/{nin,nin_gc,resnet}.py`` so that a seeded init and a
``state_dict`` are interchangeable with the reference's.
"""
from . import nin, nin_gc, resnet  # noqa: F401
