"""Import shim: the package lives in the directory ``pc-gym_amd/`` (repo layout
contract), whose name is not a valid Python identifier.  ``import pcgym_amd``
executes this file, which loads that directory as the package ``pcgym_amd`` and
replaces itself in ``sys.modules`` (sub-modules import normally afterwards)."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "pc-gym_amd")
_spec = _u.spec_from_file_location("pcgym_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["pcgym_amd"] = _mod
_spec.loader.exec_module(_mod)
