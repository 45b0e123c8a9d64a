"""Import name of the `mfvi-dip-mia_amd/` package (a hyphen cannot appear in a Python identifier)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mfvi-dip-mia_amd")
__path__.insert(0, _real)

from .api import *  # noqa: F401,F403,E402
