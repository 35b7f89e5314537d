"""Importable name of the ``real-time-reid-tracking_amd/`` package.

The product directory carries the upstream repository's name, which contains hyphens and so is not a Python identifier.
This module loads that directory AS the package ``reid_amd`` with the standard importlib machinery (a module spec whose
submodule search path is the real directory): ``import reid_amd`` returns the real package object, ``reid_amd.engine`` is
``real-time-reid-tracking_amd/engine.py``, and nothing is copied or exec'd.
"""
import importlib.util as _util
import os as _os
import sys as _sys

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "real-time-reid-tracking_amd")
_spec = _util.spec_from_file_location(__name__, _os.path.join(_REAL, "__init__.py"), submodule_search_locations=[_REAL])
_module = _util.module_from_spec(_spec)
_sys.modules[__name__] = _module          # `import reid_amd` hands out the real package from here on
_spec.loader.exec_module(_module)
