"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement (plain C + numpy) of the reference's wave-propagation hot path.
Importable only from ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg; the product package never imports it.

PARITY STATUS: pinned against the reference's own numpy helpers
(tests/golden/seisgan_helpers.npz), the analytical Green's function of
accuracy.ipynb and the Taylor-gradient criterion of gradient_example.py; the elastic
scheme against closed-form Cagniard - de Hoop solutions (full space, Garvin, Lamb:
oracle/analytic_elastic.py, tests/test_elastic_analytic_pins.py).  Parity with the
deepwave / DENISE / Devito BINARIES is unpinned (none is present).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def _sanitized():
    """ORACLE_ASAN=1: load the AddressSanitizer + UBSan builds (`make -C oracle asan`; the interpreter must have
    been started with libasan preloaded - tools/cpu_suite_asan.sh does both)."""
    return os.environ.get("ORACLE_ASAN", "0") == "1"


def build(force=False):
    """Compile the C oracle (gcc).  Building the checker is not using it."""
    sfx = "_asan" if _sanitized() else ""
    want = [os.path.join(_HERE, "liboracle_f32%s.so" % sfx), os.path.join(_HERE, "liboracle_f64%s.so" % sfx)]
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c")]
    stale = force or any(
        (not os.path.exists(w)) or any(os.path.getmtime(s) > os.path.getmtime(w) for s in srcs)
        for w in want)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "asan" if sfx else "all"], stdout=subprocess.DEVNULL)
    return want


def _cfg_struct(real):
    class Cfg(ctypes.Structure):
        _fields_ = [("n0", ctypes.c_int), ("n1", ctypes.c_int), ("nt", ctypes.c_int),
                    ("nshot", ctypes.c_int), ("nsrc", ctypes.c_int), ("nrec", ctypes.c_int),
                    ("ntap", ctypes.c_int), ("c0", real), ("c1", real)]
    return Cfg


class Oracle:
    """ctypes view of liboracle_{f32,f64}.so."""

    def __init__(self, precision="f32"):
        build()
        self.precision = precision
        self.dtype = np.float32 if precision == "f32" else np.float64
        self.creal = ctypes.c_float if precision == "f32" else ctypes.c_double
        self.lib = ctypes.CDLL(os.path.join(_HERE, "liboracle_%s%s.so" % (precision, "_asan" if _sanitized() else "")))
        assert self.lib.oracle_real_bytes() == np.dtype(self.dtype).itemsize
        self.AcCfg = _cfg_struct(self.creal)

    # -- helpers -------------------------------------------------------------------------------
    def _r(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    @staticmethod
    def _i(a):
        return np.ascontiguousarray(a, dtype=np.int32)

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(ctypes.c_void_p)

    # -- acoustic ------------------------------------------------------------------------------
    def acoustic_forward(self, r, q0, q1, f, src_cell, src_w, rec_cell, rec_w,
                         c0=1.0, c1=1.0, save=False):
        """f [nt,ns,nsrc]; src_cell/src_w [ns,nsrc,ntap]; rec_* [ns,nrec,ntap].
        Returns rec [nt,ns,nrec] (and G [nt,ns,n0,n1] when save)."""
        r = self._r(r); q0 = self._r(q0); q1 = self._r(q1); f = self._r(f)
        src_cell = self._i(src_cell); rec_cell = self._i(rec_cell)
        src_w = self._r(src_w); rec_w = self._r(rec_w)
        n0, n1 = r.shape
        nt, ns, nsrc = f.shape
        nrec, ntap = rec_cell.shape[1], rec_cell.shape[2]
        assert src_cell.shape == (ns, nsrc, ntap) and rec_cell.shape == (ns, nrec, ntap)
        cfg = self.AcCfg(n0, n1, nt, ns, nsrc, nrec, ntap, c0, c1)
        rec = np.zeros((nt, ns, nrec), dtype=self.dtype)
        G = np.zeros((nt, ns, n0, n1), dtype=self.dtype) if save else None
        st = self.lib.oracle_acoustic_forward(ctypes.byref(cfg), self._p(r), self._p(q0),
                                              self._p(q1), self._p(f), self._p(src_cell),
                                              self._p(src_w), self._p(rec_cell), self._p(rec_w),
                                              self._p(rec), self._p(G))
        if st != 0:
            raise MemoryError("oracle_acoustic_forward failed")
        return (rec, G) if save else rec

    def acoustic_born(self, r, q0, q1, dr, G, rec_cell, rec_w, c0=1.0, c1=1.0):
        """Linearised seismograms J dr [nt,ns,nrec] for a perturbation dr [n0,n1] of r, around the
        forward run that produced G [nt,ns,n0,n1]."""
        r = self._r(r); q0 = self._r(q0); q1 = self._r(q1); dr = self._r(dr); G = self._r(G)
        rec_cell = self._i(rec_cell); rec_w = self._r(rec_w)
        n0, n1 = r.shape
        nt, ns = G.shape[0], G.shape[1]
        nrec, ntap = rec_cell.shape[1], rec_cell.shape[2]
        cfg = self.AcCfg(n0, n1, nt, ns, 0, nrec, ntap, c0, c1)
        out = np.zeros((nt, ns, nrec), dtype=self.dtype)
        st = self.lib.oracle_acoustic_born(ctypes.byref(cfg), self._p(r), self._p(q0), self._p(q1),
                                           self._p(dr), self._p(G), self._p(rec_cell), self._p(rec_w),
                                           self._p(out))
        if st != 0:
            raise MemoryError("oracle_acoustic_born failed")
        return out

    def acoustic_backward(self, r, q0, q1, src_cell, src_w, rec_cell, rec_w, g, G,
                          c0=1.0, c1=1.0, want_grad_f=True):
        r = self._r(r); q0 = self._r(q0); q1 = self._r(q1); g = self._r(g); G = self._r(G)
        src_cell = self._i(src_cell); rec_cell = self._i(rec_cell)
        src_w = self._r(src_w); rec_w = self._r(rec_w)
        n0, n1 = r.shape
        nt, ns, nrec = g.shape
        nsrc, ntap = src_cell.shape[1], src_cell.shape[2]
        cfg = self.AcCfg(n0, n1, nt, ns, nsrc, nrec, ntap, c0, c1)
        grad_r = np.zeros((n0, n1), dtype=self.dtype)
        grad_f = np.zeros((nt, ns, nsrc), dtype=self.dtype) if want_grad_f else None
        st = self.lib.oracle_acoustic_backward(ctypes.byref(cfg), self._p(r), self._p(q0),
                                               self._p(q1), self._p(src_cell), self._p(src_w),
                                               self._p(rec_cell), self._p(rec_w), self._p(g),
                                               self._p(G), self._p(grad_r), self._p(grad_f))
        if st != 0:
            raise MemoryError("oracle_acoustic_backward failed")
        return grad_r, grad_f


def _acoustic_forward_order(self, order, r, q0, q1, f, src_cell, src_w, rec_cell, rec_w, c0=1.0, c1=1.0,
                            save_u=False):
    """Forward modelling with a space-order-`order` Laplacian (2..20, Devito's Taylor weights);
    returns rec [nt,ns,nrec] (and U [nt,ns,n0,n1] = u^n when save_u).  Reference runs only."""
    r = self._r(r); q0 = self._r(q0); q1 = self._r(q1); f = self._r(f)
    src_cell = self._i(src_cell); rec_cell = self._i(rec_cell)
    src_w = self._r(src_w); rec_w = self._r(rec_w)
    n0, n1 = r.shape
    nt, ns, nsrc = f.shape
    nrec, ntap = rec_cell.shape[1], rec_cell.shape[2]
    cfg = self.AcCfg(n0, n1, nt, ns, nsrc, nrec, ntap, c0, c1)
    rec = np.zeros((nt, ns, nrec), dtype=self.dtype)
    U = np.zeros((nt, ns, n0, n1), dtype=self.dtype) if save_u else None
    st = self.lib.oracle_acoustic_forward_order(ctypes.byref(cfg), int(order), self._p(r), self._p(q0),
                                                self._p(q1), self._p(f), self._p(src_cell), self._p(src_w),
                                                self._p(rec_cell), self._p(rec_w), self._p(rec), self._p(U))
    if st != 0:
        raise RuntimeError("oracle_acoustic_forward_order failed (%d)" % st)
    return (rec, U) if save_u else rec


def _acoustic_gradient_devito(self, r, q0, q1, rec_cell, rec_w, res, U_dev, s, h, c0=1.0, c1=1.0):
    """Devito's `grad -= u.dt2 * v` (operators.py:127-165) in Devito's time indexing: res [nt,ns,nrec]
    is the residual at Devito time index, U_dev [nt,ns,n0,n1] Devito's saved u[time].  Returns the
    gradient w.r.t. square slowness on the padded grid, summed over shots."""
    r = self._r(r); q0 = self._r(q0); q1 = self._r(q1); res = self._r(res); U_dev = self._r(U_dev)
    rec_cell = self._i(rec_cell); rec_w = self._r(rec_w)
    n0, n1 = r.shape
    nt, ns, nrec = res.shape
    ntap = rec_cell.shape[2]
    cfg = self.AcCfg(n0, n1, nt, ns, 0, nrec, ntap, c0, c1)
    grad = np.zeros((n0, n1), dtype=self.dtype)
    st = self.lib.oracle_acoustic_gradient_devito(ctypes.byref(cfg), self._p(r), self._p(q0), self._p(q1),
                                                  self._p(rec_cell), self._p(rec_w), self._p(res),
                                                  self._p(U_dev), self.creal(s), self.creal(h), self._p(grad))
    if st != 0:
        raise MemoryError("oracle_acoustic_gradient_devito failed")
    return grad


def _acoustic_cpml_forward(self, r, ab0, ab1, f, src_cell, src_w, rec_cell, rec_w, c0=1.0, c1=1.0, save=False):
    """oracle/acoustic_cpml.c: the scalar scheme with a second-order C-PML.  ab0 [2,n0], ab1 [2,n1] = the a and b
    profiles of the layer along each axis (zero outside it); everything else as :meth:`acoustic_forward`."""
    r = self._r(r); ab0 = self._r(ab0); ab1 = self._r(ab1); f = self._r(f)
    src_cell = self._i(src_cell); rec_cell = self._i(rec_cell)
    src_w = self._r(src_w); rec_w = self._r(rec_w)
    n0, n1 = r.shape
    nt, ns, nsrc = f.shape
    nrec, ntap = rec_cell.shape[1], rec_cell.shape[2]
    assert ab0.shape == (2, n0) and ab1.shape == (2, n1)
    cfg = self.AcCfg(n0, n1, nt, ns, nsrc, nrec, ntap, c0, c1)
    rec = np.zeros((nt, ns, nrec), dtype=self.dtype)
    G = np.zeros((nt, ns, n0, n1), dtype=self.dtype) if save else None
    st = self.lib.oracle_acoustic_cpml_forward(ctypes.byref(cfg), self._p(r), self._p(ab0), self._p(ab1), self._p(f),
                                               self._p(src_cell), self._p(src_w), self._p(rec_cell), self._p(rec_w),
                                               self._p(rec), self._p(G))
    if st != 0:
        raise MemoryError("oracle_acoustic_cpml_forward failed")
    return (rec, G) if save else rec


def _acoustic_cpml_backward(self, r, ab0, ab1, src_cell, src_w, rec_cell, rec_w, g, G, c0=1.0, c1=1.0,
                            want_grad_f=True):
    r = self._r(r); ab0 = self._r(ab0); ab1 = self._r(ab1); g = self._r(g); G = self._r(G)
    src_cell = self._i(src_cell); rec_cell = self._i(rec_cell)
    src_w = self._r(src_w); rec_w = self._r(rec_w)
    n0, n1 = r.shape
    nt, ns, nrec = g.shape
    nsrc, ntap = src_cell.shape[1], src_cell.shape[2]
    cfg = self.AcCfg(n0, n1, nt, ns, nsrc, nrec, ntap, c0, c1)
    grad_r = np.zeros((n0, n1), dtype=self.dtype)
    grad_f = np.zeros((nt, ns, nsrc), dtype=self.dtype) if want_grad_f else None
    st = self.lib.oracle_acoustic_cpml_backward(ctypes.byref(cfg), self._p(r), self._p(ab0), self._p(ab1),
                                                self._p(src_cell), self._p(src_w), self._p(rec_cell), self._p(rec_w),
                                                self._p(g), self._p(G), self._p(grad_r), self._p(grad_f))
    if st != 0:
        raise MemoryError("oracle_acoustic_cpml_backward failed")
    return grad_r, grad_f


Oracle.acoustic_cpml_forward = _acoustic_cpml_forward
Oracle.acoustic_cpml_backward = _acoustic_cpml_backward
Oracle.acoustic_forward_order = _acoustic_forward_order
Oracle.acoustic_gradient_devito = _acoustic_gradient_devito


def _el_cfg():
    class ElCfg(ctypes.Structure):
        _fields_ = [("nz", ctypes.c_int), ("nx", ctypes.c_int), ("nt", ctypes.c_int),
                    ("nshot", ctypes.c_int), ("nsrc", ctypes.c_int), ("nrec", ctypes.c_int),
                    ("ntap", ctypes.c_int), ("free_surface", ctypes.c_int),
                    ("source_type", ctypes.c_int), ("fd_order", ctypes.c_int)]
    return ElCfg


def _elastic_forward(self, mat, pz, px, f, src_cell, src_w, rec_cell, rec_w, save=False,
                     free_surface=0, source_type=0, pressure=False, fd_order=4):
    """mat [5,nz,nx]; pz [6,nz]; px [6,nx]; f [nt,ns,nsrc] -> rec_vx, rec_vz [nt,ns,nrec]
    (and S [nt,ns,5,nz,nx] when save).  source_type 0: f added to sxx and szz; 1 / 2: to vx / vz."""
    mat = self._r(mat); pz = self._r(pz); px = self._r(px); f = self._r(f)
    src_cell = self._i(src_cell); rec_cell = self._i(rec_cell)
    src_w = self._r(src_w); rec_w = self._r(rec_w)
    _, nz, nx = mat.shape
    nt, ns, nsrc = f.shape
    nrec, ntap = rec_cell.shape[1], rec_cell.shape[2]
    cfg = _el_cfg()(nz, nx, nt, ns, nsrc, nrec, ntap, free_surface, source_type, fd_order)
    rvx = np.zeros((nt, ns, nrec), dtype=self.dtype)
    rvz = np.zeros((nt, ns, nrec), dtype=self.dtype)
    S = np.zeros((nt, ns, 5, nz, nx), dtype=self.dtype) if save else None
    rp = np.zeros((nt, ns, nrec), dtype=self.dtype) if pressure else None   # sum w (sxx + szz)
    st = self.lib.oracle_elastic_forward(ctypes.byref(cfg), self._p(mat), self._p(pz), self._p(px),
                                         self._p(f), self._p(src_cell), self._p(src_w),
                                         self._p(rec_cell), self._p(rec_w), self._p(rvx),
                                         self._p(rvz), self._p(S), self._p(rp))
    if st != 0:
        raise RuntimeError("oracle_elastic_forward failed (%d)" % st)
    out = (rvx, rvz, S) if save else (rvx, rvz)
    return out + (rp,) if pressure else out


def _elastic_backward(self, mat, pz, px, src_cell, src_w, rec_cell, rec_w, g_vx, g_vz, S,
                      want_grad_f=True, free_surface=0, source_type=0, g_p=None, fd_order=4):
    mat = self._r(mat); pz = self._r(pz); px = self._r(px)
    g_vx = self._r(g_vx); g_vz = self._r(g_vz); S = self._r(S)
    g_p = None if g_p is None else self._r(g_p)
    src_cell = self._i(src_cell); rec_cell = self._i(rec_cell)
    src_w = self._r(src_w); rec_w = self._r(rec_w)
    _, nz, nx = mat.shape
    nt, ns, nrec = g_vx.shape
    nsrc, ntap = src_cell.shape[1], src_cell.shape[2]
    cfg = _el_cfg()(nz, nx, nt, ns, nsrc, nrec, ntap, free_surface, source_type, fd_order)
    gm = np.zeros((5, nz, nx), dtype=self.dtype)
    gf = np.zeros((nt, ns, nsrc), dtype=self.dtype) if want_grad_f else None
    st = self.lib.oracle_elastic_backward(ctypes.byref(cfg), self._p(mat), self._p(pz),
                                          self._p(px), self._p(src_cell), self._p(src_w),
                                          self._p(rec_cell), self._p(rec_w), self._p(g_vx),
                                          self._p(g_vz), self._p(S), self._p(gm), self._p(gf),
                                          self._p(g_p))
    if st != 0:
        raise RuntimeError("oracle_elastic_backward failed (%d)" % st)
    return gm, gf


Oracle.elastic_forward = _elastic_forward
Oracle.elastic_backward = _elastic_backward


def load(precision="f32"):
    if precision not in _LIBS:
        _LIBS[precision] = Oracle(precision)
    return _LIBS[precision]
