"""Alias of ``micronet_amd.models.resnet`` under the reference's module path (micronet/models/resnet.py)."""
import micronet_amd.models.resnet as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
