"""Importable alias for the ``real-time-reid-tracking_amd/`` package directory.

The product directory name contains hyphens (it mirrors the upstream repository
name) and therefore cannot be imported with a plain ``import`` statement.  This
shim makes ``import reid_amd`` resolve every submodule from that directory.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "real-time-reid-tracking_amd")
__path__.insert(0, _REAL)  # submodules: reid_amd.<x> -> real-time-reid-tracking_amd/<x>.py

with open(_os.path.join(_REAL, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_REAL, "__init__.py"), "exec"))
del _f
