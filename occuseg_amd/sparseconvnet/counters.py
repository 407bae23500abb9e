"""Forward-pass MAC / hidden-state counters kept as attributes of the package,
as the reference keeps them (sparseconvnet/__init__.py:7-8,
sparseconvnet/submanifoldConvolution.py:97-107): user code reads and resets
``scn.forward_pass_multiplyAdd_count`` / ``scn.forward_pass_hidden_states``."""
import sys

_PKG = __name__.rsplit(".", 1)[0]


def add(macs, hidden):
    pkg = sys.modules[_PKG]
    pkg.forward_pass_multiplyAdd_count += macs
    pkg.forward_pass_hidden_states += hidden
