"""kinetica_jl_amd - MI355X-native drop-in for Kinetica.jl's kinetic-ODE solve path.

(The directory is spelled with an underscore because `kinetica.jl_amd` is not an importable
Python package name.) Host-side mirror of the reference's src/solving interface on top of the
C ABI in include/kinetica_hip.h; all numerics run in libkinetica_hip.so (HIP, gfx950).
"""
from . import capi  # noqa: F401
