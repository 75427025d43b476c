"""astroburst_amd -- MI355X (gfx950) implementation of AstroBurst's pixel-compute hot path.

The product is astroburst_amd/libastroburst_hip.so (hand-written HIP behind the C ABI in
include/astroburst_hip.h).  `core` mirrors the reference's `core::*` Rust functions over that
ABI for tests and benchmarks; `synth` makes deterministic synthetic frame stacks.
"""
from ._lib import AstroBurstError, LIB_PATH, build, declared_symbols, is_dev_build, version  # noqa: F401
from .core import Comm, Context, ImageStats, PlaneList, StackResult, StfParams  # noqa: F401

__version__ = "0.2.0"
