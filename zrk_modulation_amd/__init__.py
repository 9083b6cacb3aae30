"""zrk_modulation_amd: the ZRK simulator's per-tick hot path (AirEnv advance, SectorRadar sweep,
detection compaction, Missile step) as HIP kernels for MI355X behind the reference's module API.

Importing the package is cheap and GPU-free; anything that computes goes through
libzrk_hot.so (zrk_modulation_amd._lib) and raises HotPathUnavailable when that is impossible.
"""
from ._lib import HotPathUnavailable, ZrkError  # noqa: F401

__all__ = ["HotPathUnavailable", "ZrkError"]
__version__ = "0.1.0"
