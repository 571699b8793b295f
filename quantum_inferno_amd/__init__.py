"""Import shim: the package lives in the directory `quantum-inferno_amd/`, which is not a valid
Python identifier.  `import quantum_inferno_amd` resolves every submodule from there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "quantum-inferno_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _fh:
    exec(compile(_fh.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _fh, _real
