"""``import physicsbasedfwi2_amd.compat.deepwave as deepwave`` -- the subset of the old
(<=0.0.9) deepwave API that models/networks.py uses (line 10 import; call sites 5357,
5408-5411, 5449, 5464 and the 21 sibling ``prop`` methods)."""
from . import scalar, wavelets  # noqa: F401
