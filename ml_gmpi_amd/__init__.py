"""Import shim: the package lives in the directory `ml-gmpi_amd/` (the name the project layout
prescribes), which is not a valid Python identifier.  `import ml_gmpi_amd` resolves its submodules
from that directory."""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "ml-gmpi_amd")
if not _os.path.isdir(_REAL):  # pragma: no cover
    raise ImportError(f"package directory not found: {_REAL}")
__path__.insert(0, _REAL)

from ._api import *  # noqa: E402,F401,F403
from ._api import __all__  # noqa: E402,F401
