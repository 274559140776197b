"""Import shim: the package directory is named ``modern-rzip_amd`` (not a valid
Python identifier), so ``import modern_rzip_amd`` resolves here and re-exports it."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "modern-rzip_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
