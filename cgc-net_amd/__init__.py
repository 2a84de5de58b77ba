"""cgc_net_amd -- MI355X-native (gfx950) hot path of CGC-Net.

The package is importable on a CPU-only host (data containers, module construction,
state_dict handling); every compute entry point goes through ``libcgc_hip.so`` and
raises if that library or a GPU is missing -- there is no CPU fallback.
"""
from . import data  # noqa: F401

__all__ = ['data']
