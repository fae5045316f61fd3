"""MI355X-native implementation of LIST's SDF query hot path (import as `list_amd`)."""
from .version import __version__  # noqa: F401
