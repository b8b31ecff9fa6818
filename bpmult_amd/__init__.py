"""Import alias for the package directory ``biprojection-multimodal-transformer_amd``.

The on-disk package name carries hyphens (it is named after the reference
repository) and so cannot be written in an ``import`` statement; this stub
points ``bpmult_amd`` at that directory, so ``import bpmult_amd.models`` loads
``biprojection-multimodal-transformer_amd/models``.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "biprojection-multimodal-transformer_amd")
__path__ = [_PKG_DIR]

from . import _lib  # noqa: E402,F401
