"""gp_ss_ak_amd -- MI355X (gfx950) implementation of the GP_SS_AK hot path.

The product is the C-ABI library libgpak_hip.so (include/gpak.h) plus the host-side mirror of
the reference's Kernels/GP_utils interface in host/ (C++).  This package is the thin Python
harness around the C-ABI used by the tests and the benchmark.
"""
from . import synth  # noqa: F401

__all__ = ["synth", "gpak", "_lib"]
