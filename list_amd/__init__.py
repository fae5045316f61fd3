"""Importable alias of the package directory `learning-implicitly-from-spatial-transformers-network_amd/`
(hyphens are not valid in a Python module name): `import list_amd.hip`, `list_amd.network.modules` ...
"""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                          "learning-implicitly-from-spatial-transformers-network_amd")]
from .version import __version__  # noqa: E402,F401
