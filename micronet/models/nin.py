"""Alias of ``micronet_amd.models.nin`` under the reference's module path (micronet/models/nin.py)."""
import micronet_amd.models.nin as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
