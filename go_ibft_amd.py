"""Import alias: the package directory is ``go-ibft_amd/`` (not a valid Python
identifier), so ``import go_ibft_amd`` resolves to it through this shim."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "go-ibft_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
