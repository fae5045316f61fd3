"""Synthetic inputs of the parity cases (TEST INFRASTRUCTURE): the generators themselves live in
list_amd/synthetic.py (the bench's GPU leg uses them without importing oracle/); this module re-exports
them under the name the golden generator and the tests have always used."""
from list_amd.synthetic import *                                       # noqa: F401,F403
from list_amd.synthetic import IMG_CHANNELS, VOX_CHANNELS, _bits, _mix  # noqa: F401
