"""occuseg_amd -- MI355X-native sparse 3D submanifold-convolution backbone.

The product is `libscn_hip.so` (hand-written HIP for gfx950 behind the C ABI of
include/scn_hip.h) plus the host-side mirror of the reference's operator
interface: `occuseg_amd.SCN` (the `sparseconvnet.SCN` extension boundary) and
`occuseg_amd.sparseconvnet` (the module classes the ScanNet UNet uses).
"""
import sys

__version__ = "0.1.0"


def install():
    """Register the mirror as top-level `sparseconvnet` so reference model code
    (`import sparseconvnet as scn`, examples/ScanNet/model.py:1) runs unchanged."""
    from . import sparseconvnet as pkg, SCN
    sys.modules.setdefault("sparseconvnet", pkg)
    sys.modules.setdefault("sparseconvnet.SCN", SCN)
    return pkg
