from micronet_amd.base_module.op import *  # noqa: F401,F403
from micronet_amd.base_module.op import Add  # noqa: F401
