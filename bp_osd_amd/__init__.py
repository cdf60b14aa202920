"""bp_osd_amd -- MI355X (gfx950) native BP+OSD decoder, drop-in for the decode path of
quantumgizmos/bp_osd (`bposd_decoder(...).decode(syndrome)`).

    from bp_osd_amd import bposd_decoder        # instead of: from ldpc import bposd_decoder
    from bp_osd_amd import BpOsdDecoder         # instead of: from ldpc import BpOsdDecoder

Importing the package does not touch the GPU; constructing a decoder does, and fails loudly
when the HIP library or a device is missing (there is no CPU fallback).
"""
from .decoder import BpOsdDecoder, bposd_decoder  # noqa: F401
from . import codes  # noqa: F401

__version__ = "0.1.0"


def get_include():
    """Directory of the C-ABI header (mirrors bposd.get_include(), src/bposd/__init__.py:6-8)."""
    import os

    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
