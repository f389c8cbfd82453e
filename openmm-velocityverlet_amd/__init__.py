"""MI355X-native (HIP/gfx950) backend for the z-gong/openmm-velocityVerlet integrator hot path.

Import with ``importlib.import_module("openmm-velocityverlet_amd")`` (the directory name the
build contract fixes contains a hyphen).  Nothing here falls back to a CPU path: everything that
computes goes through ``lib/libvvhip.so`` (hand-written HIP behind the C ABI of include/vvhip.h).
"""
from . import systems  # noqa: F401
