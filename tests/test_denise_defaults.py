"""Defaults of the pyapi_denise-shaped shim a drop-in caller inherits without setting them (CPU: no propagation)."""


def test_fd_order_default_is_the_upstream_one():
    """The reference never assigns FD_ORDER (`#d.FD_ORDER = 4`, models/networks.py:10447, is commented out), so its
    prop() variants run pyapi_denise's default, 2; 4 is honoured on request, others raise."""
    import physicsbasedfwi2_amd.compat.pyapi_denise as api
    d = api.Denise("/nonexistent", verbose=0)
    assert d.FD_ORDER == 2
    assert d.PHYSICS == 1 and d.QUELLART == 1 and d.QUELLTYP == 1 and d.SEISMO == 1


def test_assignments_are_classified_not_swallowed():
    """`d.NAME = value` on the shim: served parameters are stored, DENISE parameters without effect on one gradient
    evaluation are stored with ONE warning, parameters that would change the result accept their neutral value only,
    anything else (a typo) raises - the reference configures DENISE by assignment alone (networks.py:11011-11043)."""
    import warnings
    import pytest
    import physicsbasedfwi2_amd.compat.pyapi_denise as api
    d = api.Denise(None, 0)
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # the constructor's own defaults warn about nothing
        d2 = api.Denise(None, 0)
        d2.PHYSICS, d2.ITERMAX, d2.TIME, d2.INVMAT1, d2.FD_ORDER = 1, 1, 5.0, 2, 4
        d2.fwi_stages = []
        d2.add_fwi_stage(fc_low=0.0, fc_high=5.0)
    assert d2.INVMAT1 == 2 and d2.NPROCX == 1
    api._warned.discard("NPROCX")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        d.NPROCX = 6
        d.NPROCX = 5                                        # once per name
    assert d.NPROCX == 5 and len([x for x in w if "NPROCX" in str(x.message)]) == 1
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d.VPUPPERLIM, d.SEIS_FILE_VX, d.JACOBIAN = 4509.0, "su/seis_x.su", "jacobian/jacobian_Test"
    assert d.VPUPPERLIM == 4509.0
    d.TIMEWIN = 0
    with pytest.raises(api.MifwiError):
        d.TIMEWIN = 1
    with pytest.raises(api.MifwiError):
        d.SWS_TAPER_GRAD_SOURCES = 1
    for typo in ("INVMAT_1", "FREESURF", "Physics", "QUELART"):
        with pytest.raises(AttributeError):
            setattr(d, typo, 1)
    d._private_note = 3                                     # the shim's own state


def test_deepwave_shim_defaults_to_a_pml_of_the_references_width_and_never_runs_on_the_cpu():
    """`deepwave.scalar.Propagator({'vp': model}, dx)` as models/networks.py:3360 constructs it (no pml_width: deepwave's
    own default, 10 cells each side - SURVEY.md appendix C) gets a PML, not the sponge; the sponge is an explicit
    opt-out; unknown modes and CPU models fail loudly (the product has no CPU path)."""
    import pytest
    import torch
    from physicsbasedfwi2_amd import MifwiError
    from physicsbasedfwi2_amd.compat.deepwave import scalar
    vp = torch.full((20, 30), 2000.0)
    p = scalar.Propagator({"vp": vp}, 10.0)
    assert p.absorbing == "cpml" and p.pml_width == scalar.DEFAULT_PML_WIDTH == 10 and tuple(p.spacing) == (10.0, 10.0)
    assert scalar.Propagator({"vp": vp}, 10.0, pml_width=25, absorbing="sponge").pml_width == 25
    with pytest.raises(MifwiError):
        scalar.Propagator({"vp": vp}, 10.0, absorbing="none")
    with pytest.raises(MifwiError):
        scalar.Propagator({"rho": vp}, 10.0)
    with pytest.raises(MifwiError, match="no CPU fallback"):
        p(torch.zeros(8, 1, 1), torch.zeros(1, 1, 2), torch.zeros(1, 1, 2), 1e-3)
