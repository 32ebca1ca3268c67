"""seisgan/fwi-shaped boundary: ``FWIConfiguration`` and ``FWILoss`` of the reference's
seisgan/fwi/layers.py (67-142, 145-200), served by the HIP acoustic propagator instead of Devito.

    cfg  = FWIConfiguration(config_dict, ground_truth_m)     # m = square slowness [nx, nz]
    loss = FWILoss(cfg)(x)                                   # x [1,1,nx,nz] square slowness
    loss.backward()                                          # x.grad = grad / max|grad|

Kept from the reference: units (m, ms, km/s, kHz), axis order (x, z), edge padding by ``nbpml``,
the sponge of model.py:6-29, dt = 0.42 min(h)/max(vp) (model.py:160-168), TimeAxis arithmetic,
the Ricker with its 2/f0 delay (source.py:230), source line of layers.py:81-89 (single shot: the
literal ``int(nx/2)`` metres of line 89), receiver line of layers.py:137-142, bilinear sparse
operators with offset nbpml, Devito's time loop (time = 1..nt-2: src[time] -> u[time+1],
rec[time] <- u[time]), objective 0.5||syn-obs||^2 summed over shots, gradient cropped to the
physical domain and max-normalised in backward (layers.py:185-197).
The model gradient is the exact discrete adjoint of the forward recursion, which is what Devito's
``grad -= u.dt2*v`` (operators.py:152-153) evaluates: the two agree to round-off on the whole padded grid
(tests/test_reference_pins.py::test_devito_imaging_condition_is_the_exact_discrete_adjoint, fp64, 1e-11).
"""
import numpy as np
import torch

from .. import acoustic, misfit, profiles
from .._lib import MifwiError


class FWIConfiguration(object):
    def __init__(self, config, ground_truth_vp, device=None):
        self.config = config
        self.dtype = np.float32
        self.device = torch.device(device) if device is not None else \
            torch.device("cuda", torch.cuda.current_device())
        c = config
        self.shape = tuple(int(v) for v in c["shape"])
        self.spacing = tuple(float(v) for v in c["spacing"])
        self.nbpml = int(c["nbpml"])
        m_true = torch.as_tensor(np.asarray(ground_truth_vp, dtype=np.float32))
        if tuple(m_true.shape) != self.shape:
            raise MifwiError("ground truth %s does not match config shape %s"
                             % (tuple(m_true.shape), self.shape))
        nshots = int(c["nshots"])
        self.source_locations = np.empty((nshots, 2), dtype=np.float32)
        self.source_locations[:, 1] = c["source_min_y"]
        self.source_locations[:, 0] = np.linspace(c["source_min_x"],
                                                  self.spacing[0] * self.shape[0] - c["source_min_x"],
                                                  num=nshots)
        if nshots == 1:
            self.source_locations[:, 0] = np.array([int(self.shape[0] / 2.)])
        vp_max = float(1.0 / np.sqrt(float(m_true.min())))
        self.dt = profiles.seisgan_critical_dt(self.spacing, vp_max)
        self.nt, self.stop = profiles.time_axis(0.0, float(c["tn"]), self.dt)
        self.time_values = np.linspace(0.0, self.stop, self.nt)
        self.wavelet = profiles.ricker_seisgan(float(c["f0"]), self.time_values).astype(np.float32)
        nrec = int(c["nreceivers"])
        self.rec_coords = np.zeros((nrec, 2), dtype=np.float32)
        self.rec_coords[:, 1] = c["rec_min_y"]
        self.rec_coords[:, 0] = np.linspace(0, nrec * self.spacing[0], num=nrec)
        self._build_static()
        with torch.no_grad():
            clean = self.model(m_true.to(self.device))
        self.clean_ds = clean.sum(dim=1).cpu().numpy()
        self.noise_norm = 0.0
        rng = np.random
        noisy = clean.clone()
        for i in range(nshots):
            d = clean[:, i, :]
            noise = float(c.get("noise_percent", 0.0)) * float(d.std()) * rng.randn(*d.shape)
            self.noise_norm += 0.5 * np.linalg.norm(noise) ** 2
            noisy[:, i, :] += torch.as_tensor(noise, dtype=torch.float32, device=self.device)
        self.true_ds = noisy                      # [nt, nshots, nrec]
        self.noisy_ds = noisy.sum(dim=1).cpu().numpy()

    def _build_static(self):
        p, h = self.nbpml, self.spacing
        n0, n1 = self.shape[0] + 2 * p, self.shape[1] + 2 * p
        self.shape_pml = (n0, n1)
        href = min(h)
        self.c0, self.c1 = (href / h[0]) ** 2, (href / h[1]) ** 2
        self.q0 = torch.from_numpy(profiles.sponge_q(n0, p, h[0], href, self.dt)).float()
        self.q1 = torch.from_numpy(profiles.sponge_q(n1, p, h[1], href, self.dt)).float()
        ns = self.source_locations.shape[0]
        self.sc, self.sw = profiles.cells_bilinear(self.source_locations[:, None, :], h, p, (n0, n1))
        rec = np.broadcast_to(self.rec_coords[None], (ns,) + self.rec_coords.shape)
        self.rc, self.rw = profiles.cells_bilinear(rec, h, p, (n0, n1))
        # Devito loop: my f[n] = src[n+1] for n = 0..nt-3 ; source term  w src s^2/m = w f r, f = src h^2
        f = np.zeros((self.nt, ns, 1), dtype=np.float32)
        f[:self.nt - 2, :, 0] = (self.wavelet[1:self.nt - 1] * href * href)[:, None]
        self.f = torch.from_numpy(f).to(self.device)

    def model(self, m):
        """Seismograms [nt, nshots, nrec] for square slowness m [nx, nz] (differentiable)."""
        p, href = self.nbpml, min(self.spacing)
        m_pad = torch.nn.functional.pad(m[None, None].float(), (p, p, p, p), mode="replicate")[0, 0]
        r = (self.dt * self.dt / (href * href)) / m_pad
        rec = acoustic.propagate(r, self.f, self.q0, self.q1, self.sc, self.sw, self.rc, self.rw,
                                 self.c0, self.c1)
        out = torch.zeros_like(rec)
        out[1:self.nt - 1] = rec[0:self.nt - 2]          # rec_devito[t] = rec[t-1]
        return out


class _FWILossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, cfg, holder):
        with torch.enable_grad():
            m = x[0, 0].detach().to(cfg.device).requires_grad_(True)
            p, href = cfg.nbpml, min(cfg.spacing)
            # the reference crops the gradient of the PADDED array (layers.py:185-186): keep the
            # padded square slowness as the differentiation variable and crop afterwards
            m_pad = torch.nn.functional.pad(m[None, None].float(), (p, p, p, p), mode="replicate")[0, 0]
            m_pad = m_pad.detach().requires_grad_(True)
            r = (cfg.dt * cfg.dt / (href * href)) / m_pad
            rec = acoustic.propagate(r, cfg.f, cfg.q0, cfg.q1, cfg.sc, cfg.sw, cfg.rc, cfg.rw,
                                     cfg.c0, cfg.c1)
            syn = torch.zeros_like(rec)
            syn[1:cfg.nt - 1] = rec[0:cfg.nt - 2]
            objective = misfit.l2_half(syn, cfg.true_ds)      # layers.py:176-178, fused on the GPU
            (g_pad,) = torch.autograd.grad(objective, m_pad)
        gradient = g_pad[p:-p, p:-p] if p > 0 else g_pad
        holder.smooth_ds = syn.detach().sum(dim=1).cpu().numpy()
        holder.gradient = gradient.detach()
        ctx.save_for_backward(gradient.detach())
        ctx.x_device = x.device
        return objective.detach().reshape(1).float().to(x.device)

    @staticmethod
    def backward(ctx, grad_output):
        (gradient,) = ctx.saved_tensors
        grad_input = (gradient / gradient.abs().max()).unsqueeze(0).unsqueeze(0)
        return grad_input.to(ctx.x_device), None, None


class FWILoss(object):
    """Callable like the reference's legacy autograd.Function instance: ``FWILoss(cfg)(x)``."""

    def __init__(self, configuration):
        self.config = configuration
        self.gradient = None
        self.smooth_ds = None

    def __call__(self, x):
        return _FWILossFn.apply(x, self.config, self)

    forward = __call__

    def reset(self):
        self.smooth_ds = None
