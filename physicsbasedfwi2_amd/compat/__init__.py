"""Drop-in shims reproducing the three propagator call protocols of the reference."""
