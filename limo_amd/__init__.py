"""limo_amd — MI355X-native hot path of LIMO's keyframe bundle adjustment and LiDAR depth assignment.

The product is the C-ABI shared library `limo_amd/lib/liblimo_hip.so` (hand-written HIP for gfx950, see
include/limo_hip.h) plus the C++ host shim under `limo_amd/kba/` that keeps the reference's
`keyframe_bundle_adjustment` class surface.  This Python package is plumbing for tests and the benchmark:
ctypes bindings (`_ffi`), a numpy window container (`window`) and the synthetic KITTI-shaped generator (`synth`).
"""
from . import _ffi  # noqa: F401
from .window import Window, default_options, struct_array  # noqa: F401
