"""Alias of ``micronet_amd.quantization.wqaq.iao.quantize`` under the reference's module path
(micronet/compression/quantization/wqaq/iao/quantize.py); ``import quantize`` from a reference ``main.py`` resolves
here when this directory is the script's cwd."""
import micronet_amd.quantization.wqaq.iao.quantize as _impl
from micronet_amd.quantization.wqaq.iao.quantize import *  # noqa: F401,F403

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
