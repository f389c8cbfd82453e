"""MI355X-native (HIP/gfx950) backend for the z-gong/openmm-velocityVerlet integrator hot path.

Import with ``importlib.import_module("openmm-velocityverlet_amd")`` (the directory name the build
contract fixes contains a hyphen).  ``systems`` is plain numpy; ``vvhip`` / ``integrator`` load
``lib/libvvhip.so`` (hand-written HIP behind the C ABI of include/vvhip.h) and fail loudly if it has
not been built -- there is no CPU or PyTorch fallback for the hot path.
"""
import importlib as _importlib

from . import systems  # noqa: F401

_LAZY = ("vvhip", "integrator", "distributed")


def __getattr__(name):
    if name in _LAZY:
        return _importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)
