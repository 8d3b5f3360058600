from micronet_amd.models import nin, nin_gc, resnet  # noqa: F401
