"""deepwave.wavelets as used at models/networks.py:5357."""
from ...profiles import ricker  # noqa: F401
