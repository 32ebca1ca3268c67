"""physicsbasedfwi2_amd -- MI355X-native (gfx950, hand-written HIP) 2-D finite-difference
wave propagator behind the call protocols the PhysicsBasedFWI2 training loop uses
(deepwave-shaped acoustic, pyapi_denise-shaped elastic, seisgan/Devito-shaped FWILoss).

Only the propagation hot path lives here; see DESIGN.md for scope and INTEGRATION.md for the
one-line import changes on the reference side.
"""
from ._lib import MifwiError  # noqa: F401

__version__ = "0.1.0"
