"""The single-launch time loops hand halo rows between workgroups and give up (bounded spins) when
a neighbour never shows up.  The call then re-runs with one launch per (half) step instead of failing:
from the zero state when it started there, from a copy of its input state when it was a resumed
(time-checkpointed) call.  MIFWI_TEST_FAKE_TIMEOUT=1 makes the host treat every single-launch attempt as
timed out before it runs, =2 after it has run (the state really has to be restored); the device side of
the time-out (a workgroup that never shows up) is driven in an ablation build, in a child process."""
import os
import subprocess
import sys


import numpy as np
import pytest
import torch

from cases import acoustic_case, elastic_case, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(case, *names):
    return [torch.tensor(case[n]) for n in names]


def _acoustic(case, **kw):
    from physicsbasedfwi2_amd import acoustic
    r = torch.tensor(case["r"], dtype=torch.float32, device=DEV, requires_grad=True)
    f = torch.tensor(case["f"], dtype=torch.float32, device=DEV, requires_grad=True)
    rec = acoustic.propagate(r, f, *_t(case, "q0", "q1", "sc", "sw", "rc", "rw"), case["c0"], case["c1"], **kw)
    rec.backward(torch.sign(rec.detach()))
    return rec.detach(), r.grad, f.grad


def _elastic(case, **kw):
    from physicsbasedfwi2_amd import elastic
    mat = torch.tensor(case["mat"], dtype=torch.float32, device=DEV, requires_grad=True)
    f = torch.tensor(case["f"], dtype=torch.float32, device=DEV, requires_grad=True)
    rvx, rvz = elastic.propagate(mat, f, *_t(case, "pz", "px", "sc", "sw", "rc", "rw"), case["fw"], **kw)
    torch.autograd.backward([rvx, rvz], [torch.sign(rvx.detach()), torch.sign(rvz.detach())])
    return rvx.detach(), rvz.detach(), mat.grad, f.grad


def _same(a, b):
    for x, y in zip(a, b):
        if x.dtype == torch.float32 and x.shape == y.shape and torch.equal(x, y):
            continue
        assert rel_l2(x.cpu().numpy(), y.cpu().numpy()) <= 2e-5


@pytest.mark.expects_fallback
def test_acoustic_falls_back_to_one_launch_per_step(monkeypatch):
    from physicsbasedfwi2_amd.acoustic import AcousticPlan
    case = acoustic_case(seed=71, n0=120, n1=200, nb=10, nt=120, ns=6, nrec=40)
    n0, n1 = case["r"].shape[-2:]
    assert AcousticPlan(n0, n1, 120, 6, 1, 40, 1, case["c0"], case["c1"], 0).cluster_slabs() >= 1
    ref = _acoustic(case)
    monkeypatch.setenv("MIFWI_TEST_FAKE_TIMEOUT", "1")
    got = _acoustic(case)
    assert float(ref[0].abs().max()) > 0 and torch.equal(ref[0], got[0])     # traces bit for bit
    _same(ref[1:], got[1:])


@pytest.mark.expects_fallback
def test_acoustic_born_falls_back(monkeypatch):
    from physicsbasedfwi2_amd import acoustic
    case = acoustic_case(seed=73, n0=100, n1=160, nb=10, nt=100, ns=3, nrec=30)
    r = torch.tensor(case["r"], dtype=torch.float32, device=DEV)
    f = torch.tensor(case["f"], dtype=torch.float32, device=DEV)
    dr = 0.01 * torch.randn_like(r)
    args = (r, f, dr, *_t(case, "q0", "q1", "sc", "sw", "rc", "rw"), case["c0"], case["c1"])
    ref = acoustic.born(*args)
    monkeypatch.setenv("MIFWI_TEST_FAKE_TIMEOUT", "1")
    got = acoustic.born(*args)
    ref, got = (ref if isinstance(ref, tuple) else (ref,)), (got if isinstance(got, tuple) else (got,))
    for a, b in zip(ref, got):
        assert float(a.abs().max()) > 0 and torch.equal(a, b)


@pytest.mark.expects_fallback
def test_elastic_falls_back_to_one_launch_per_half_step(monkeypatch):
    from physicsbasedfwi2_amd.elastic import ElasticPlan
    case = elastic_case(seed=79, nz=100, nx=300, fw=10, ns=6, nrec=100, nt=120)
    pl = ElasticPlan(100, 300, 120, 6, 1, 100, 1, 10, 0)
    assert pl.cluster_slabs(False) >= 1 and pl.cluster_slabs(True) >= 1
    from physicsbasedfwi2_amd import _lib
    n0 = _lib.load().mifwi_fallback_count()
    ref = _elastic(case)
    assert _lib.load().mifwi_fallback_count() == n0          # what conftest's _no_silent_fallback watches
    monkeypatch.setenv("MIFWI_TEST_FAKE_TIMEOUT", "1")
    got = _elastic(case)
    assert _lib.load().mifwi_fallback_count() >= n0 + 2      # forward and adjoint time loops both gave up
    assert float(ref[0].abs().max()) > 0
    assert torch.equal(ref[0], got[0]) and torch.equal(ref[1], got[1])
    _same(ref[2:], got[2:])


@pytest.mark.expects_fallback
@pytest.mark.parametrize("mode", ["1", "2"])
def test_resumed_calls_fall_back_too(monkeypatch, mode):
    """A checkpointed run resumes from saved state.  The library keeps a copy of a resumed call's input state, so
    a timed-out attempt - even one that has already advanced (mode 2: every attempt runs, then counts as timed
    out) - is rolled back and the range runs on the per-step kernels: same results, no error."""
    ca = acoustic_case(seed=83, n0=100, n1=160, nb=10, nt=120, ns=3, nrec=30)
    ce = elastic_case(seed=89, nz=100, nx=300, fw=10, ns=3, nrec=50, nt=120)
    ref_a = _acoustic(ca, snapshot_budget=1 << 20)            # forces several time segments
    ref_e = _elastic(ce, snapshot_budget=1 << 21)
    _same(ref_a, _acoustic(ca))
    _same(ref_e, _elastic(ce))
    monkeypatch.setenv("MIFWI_TEST_FAKE_TIMEOUT", mode)
    got_a = _acoustic(ca, snapshot_budget=1 << 20)
    got_e = _elastic(ce, snapshot_budget=1 << 21)
    assert float(ref_a[0].abs().max()) > 0 and torch.equal(ref_a[0], got_a[0])
    assert torch.equal(ref_e[0], got_e[0]) and torch.equal(ref_e[1], got_e[1])
    _same(ref_a[1:], got_a[1:])
    _same(ref_e[2:], got_e[2:])


@pytest.mark.expects_fallback
@pytest.mark.parametrize("mode", ["1", "2"])
def test_cpml_calls_fall_back_too(monkeypatch, mode):
    """The same with the second-order C-PML (the deepwave-shaped API's default layer): the state a rolled-back attempt
    restores includes the layer's memory variables - which an edge slab in the own-group form keeps in an LDS plane during
    the launch and writes back at its end - and the per-step kernels with their thin layer launches take over: resident
    and time-checkpointed calls, traces bit for bit."""
    from physicsbasedfwi2_amd import acoustic
    from test_acoustic_gpu import _cpml_case

    def run(c, **kw):
        r = torch.tensor(c["r"], dtype=torch.float32, device=DEV, requires_grad=True)
        f = torch.tensor(c["f"], dtype=torch.float32, device=DEV, requires_grad=True)
        rec = acoustic.propagate(r, f, *_t(c, "ab0", "ab1", "sc", "sw", "rc", "rw"), c["c0"], c["c1"], cpml_width=c["w"], **kw)
        rec.backward(torch.sign(rec.detach()))
        return rec.detach(), r.grad, f.grad

    c = _cpml_case(seed=97, n0=100, n1=150, w=10, nt=130, ns=3, nrec=30)
    ref = run(c)
    ref_ck = run(c, snapshot_budget=1 << 20)
    _same(ref, ref_ck)
    monkeypatch.setenv("MIFWI_TEST_FAKE_TIMEOUT", mode)
    got, got_ck = run(c), run(c, snapshot_budget=1 << 20)
    assert float(ref[0].abs().max()) > 0 and torch.equal(ref[0], got[0]) and torch.equal(ref[0], got_ck[0])
    _same(ref[1:], got[1:])
    _same(ref[1:], got_ck[1:])


_CHILD = r"""
import sys, json
import numpy as np, torch
sys.path.insert(0, %(tests)r); sys.path.insert(0, %(root)r)
from cases import acoustic_case, elastic_case
from physicsbasedfwi2_amd import acoustic, elastic
dev = "cuda:0"
t = lambda c, *n: [torch.tensor(c[k]) for k in n]
ca = acoustic_case(seed=71, n0=120, n1=200, nb=10, nt=100, ns=6, nrec=40)
r = torch.tensor(ca["r"], dtype=torch.float32, device=dev, requires_grad=True)
f = torch.tensor(ca["f"], dtype=torch.float32, device=dev)
rec = acoustic.propagate(r, f, *t(ca, "q0", "q1", "sc", "sw", "rc", "rw"), ca["c0"], ca["c1"])
rec.backward(torch.sign(rec.detach()))
# the same acoustic set-up with a 10-cell C-PML instead of the sponge (edge slabs in the own-group form)
from oracle import helpers as H
N0, N1 = ca["shape"]
ab0 = H.cpml_profiles(N0, 10, 10.0, ca["s"], float(ca["vp"].max()), 0.02)[:2]
ab1 = H.cpml_profiles(N1, 10, 10.0, ca["s"], float(ca["vp"].max()), 0.02)[:2]
rc_ = torch.tensor(ca["r"], dtype=torch.float32, device=dev, requires_grad=True)
recc = acoustic.propagate(rc_, f, torch.tensor(ab0), torch.tensor(ab1), *t(ca, "sc", "sw", "rc", "rw"), ca["c0"], ca["c1"],
                          cpml_width=10)
recc.backward(torch.sign(recc.detach()))
ce = elastic_case(seed=79, nz=100, nx=300, fw=10, ns=6, nrec=100, nt=100)
mat = torch.tensor(ce["mat"], dtype=torch.float32, device=dev, requires_grad=True)
ef = torch.tensor(ce["f"], dtype=torch.float32, device=dev)
vx, vz = elastic.propagate(mat, ef, *t(ce, "pz", "px", "sc", "sw", "rc", "rw"), ce["fw"])
torch.autograd.backward([vx, vz], [torch.sign(vx.detach()), torch.sign(vz.detach())])
from physicsbasedfwi2_amd import _lib
lib = _lib.load()
np.savez(sys.argv[1], rec=rec.detach().cpu().numpy(), gr=r.grad.cpu().numpy(), recc=recc.detach().cpu().numpy(),
         grc=rc_.grad.cpu().numpy(), vx=vx.detach().cpu().numpy(),
         vz=vz.detach().cpu().numpy(), gm=mat.grad.cpu().numpy(),
         counts=np.array([lib.mifwi_fallback_count(), lib.mifwi_agent_handoff_count(), lib.mifwi_slow_handoff_count()]))
"""


@pytest.mark.parametrize("bit", ["64", "128", "4096"])
def test_a_workgroup_that_never_shows_up_times_out_on_the_device_and_the_call_falls_back(tmp_path, bit):
    """The device side of the time-out, once: in the ablation build (libmifwi_ablations.so, built by
    __graft_entry__.build() with -DMIFWI_ABLATIONS) debug bit 64 makes slab 1 of the first shot of every
    single-launch kernel exit at once.  Its neighbours spin out, the first to give up publishes the error word, the
    rest of the launch bails within 256 polls, the host zeroes the state and runs the per-step kernels: same
    traces bit for bit, gradients to summation order.
    Bit 128: every workgroup reports a different XCD.  The hand-off granules are published with stores that stay in
    the XCD's L2, so the slabs of a shot must share an XCD; the placement check at the head of the kernels
    (mifwi::same_xcd) sees the mismatch and the launch bails out at once; the host then repeats it with granules published
    through the fabric (the AG kernel variants: correct on any placement) - still ONE launch per time loop, no fall-back
    to the per-step kernels, the same bits (mifwi_agent_handoff_count() > 0, mifwi_fallback_count() == 0).
    Bit 4096: slab 1 of the first shot stalls ~10 ms at every 64th step - late, not absent, as on a GPU shared with another
    process.  Nobody times out, results are the same, and mifwi_slow_handoff_count() says what happened."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "physicsbasedfwi2_amd", "libmifwi_ablations.so")
    if not os.path.exists(lib):
        pytest.skip("ablation build not present (python __graft_entry__.py builds it)")
    script = tmp_path / "child.py"
    script.write_text(_CHILD % {"tests": os.path.join(root, "tests"), "root": root})
    outs = []
    for tag, env in (("ref", {}), ("abl", {"MIFWI_LIB": lib, "MIFWI_AC_CL_DBG": bit, "MIFWI_EL_CL_DBG": bit})):
        out = tmp_path / (tag + ".npz")
        res = subprocess.run([sys.executable, str(script), str(out)], env=dict(os.environ, **env), capture_output=True,
                             text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        outs.append(dict(np.load(out)))
    ref, abl = outs
    fb, ag, slow = (int(v) for v in abl["counts"])
    assert [int(v) for v in ref["counts"]] == [0, 0, 0]
    if bit == "64":
        assert fb >= 4 and ag == 0                  # acoustic + elastic, forward + adjoint
    elif bit == "128":
        assert fb == 0 and ag >= 4 and slow == 0
    else:
        assert fb == 0 and ag == 0 and slow >= 4
    assert np.abs(ref["rec"]).max() > 0 and np.array_equal(ref["rec"], abl["rec"])
    assert np.abs(ref["recc"]).max() > 0 and np.array_equal(ref["recc"], abl["recc"]) and rel_l2(abl["grc"], ref["grc"]) <= 2e-5
    assert np.array_equal(ref["vx"], abl["vx"]) and np.array_equal(ref["vz"], abl["vz"])
    assert rel_l2(abl["gr"], ref["gr"]) <= 2e-5 and rel_l2(abl["gm"], ref["gm"]) <= 2e-5


def test_a_healthy_run_stays_on_the_single_launch_kernels(tmp_path):
    """The fall-back is silent in its results by design, so a regression that made every launch bail out (say, a
    placement check that never passes on this GPU) would only show as lost speed.  The library says so once per process
    on stderr: a plain run of both physics must not."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "child.py"
    script.write_text(_CHILD % {"tests": os.path.join(root, "tests"), "root": root})
    env = {k: v for k, v in os.environ.items() if k not in ("MIFWI_QUIET", "MIFWI_LIB", "MIFWI_TEST_FAKE_TIMEOUT")}
    res = subprocess.run([sys.executable, str(script), str(tmp_path / "ok.npz")], env=env, capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "falling back" not in res.stderr, res.stderr[-2000:]
    # and the note does appear when a launch is made to give up
    res = subprocess.run([sys.executable, str(script), str(tmp_path / "fb.npz")],
                         env=dict(env, MIFWI_TEST_FAKE_TIMEOUT="1"), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "falling back" in res.stderr
