"""Alias of ``micronet_amd.models.nin_gc`` under the reference's module path (micronet/models/nin_gc.py)."""
import micronet_amd.models.nin_gc as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
