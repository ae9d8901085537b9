"""Import alias: the product package lives in ``scale-mixtures-of-neural-network-gaussian-processes_amd/``
(not a valid Python identifier), so ``import smnngp`` maps onto that directory."""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "scale-mixtures-of-neural-network-gaussian-processes_amd")
__path__.insert(0, _PKG_DIR)
with open(_os.path.join(_PKG_DIR, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, "__init__.py"), "exec"))
del _f
