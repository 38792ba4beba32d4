"""Import alias: ``airfoil_cfd_tool_amd`` -> the package directory
``airfoil-cfd-tool_amd/`` (a hyphen is not importable).  Holds no code of its own."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "airfoil-cfd-tool_amd")
__path__ = [_real]
_init = _os.path.join(_real, "__init__.py")
with open(_init, "r", encoding="utf-8") as _fh:
    exec(compile(_fh.read(), _init, "exec"))
del _fh, _init
