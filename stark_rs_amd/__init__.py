"""stark_rs_amd -- MI355X-native engine behind stark-rs's univariate / trace / fri / merkle API.

Product code: HIP kernels (csrc/*.hip -> build/libstarkmi.so) reached through the C ABI in
include/stark_mi.h, plus a thin host-side mirror of the reference's types.  No CPU fallback.
"""
from ._lib import StarkMiError, build, declared_symbols  # noqa: F401
from .engine import Engine, DeviceTree, default_engine, P_REF, G_REF, P2, G2  # noqa: F401
