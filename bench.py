#!/usr/bin/env python3
"""Benchmark of the FWI hot path (forward modelling -> misfit -> adjoint -> model gradient).

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A "step" is one full gradient evaluation over the rank's shots.  value = interior
cells x time steps x shots (all ranks) / wall second, in Mcells*steps/s (BASELINE.json metric).
Shots are independent => weak scaling: every rank runs the workload's per-GPU shot count and
the only collective is one all-reduce of the model gradient (RCCL) inside the timed step.
Inputs are synthetic and resident in HBM before timing starts.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def sized_cpu_sample(run, nt0, nt_full, budget_s, bytes_per_step, mem_cap=6e9):
    """Time `run(nt)` on a sample sized to about `budget_s` seconds of CPU work: calibrate on nt0
    steps, then repeat a pass of up to the full step count (capped by snapshot memory) until the
    budget is used.  Returns (steps per pass, passes, seconds)."""
    el0 = run(nt0)
    nt = int(nt0 * budget_s / max(el0, 1e-3))
    nt = max(nt0, min(nt, nt_full, int(mem_cap / bytes_per_step)))
    reps = max(1, min(200, int(round(budget_s / max(el0 * nt / nt0, 1e-3)))))
    el = 0.0
    for _ in range(reps):
        el += run(nt)
    return nt, reps, el


def synth_vp(nz, nx, seed, water_rows=26):
    """1500 + 2500 z/nz + Gaussian-smoothed (sigma=5) N(0,150^2), clipped to
    [1500,4500], water layer on top."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    z = np.arange(nz, dtype=np.float64)[:, None] / nz
    vp = 1500.0 + 2500.0 * z + gaussian_filter(rng.normal(0.0, 150.0, (nz, nx)), 5.0) * 5.0
    vp = np.clip(vp, 1500.0, 4500.0)
    vp[:water_rows, :] = 1500.0
    return vp.astype(np.float32)


class AcousticMarmousi:
    """BASELINE.json configs[1]: 2-D acoustic Marmousi-like 174x500, 29 shots, 2000 steps,
    deepwave-shaped call protocol + the L1 trace-normalised misfit of networks.py:5467-5476."""
    name = "acoustic_marmousi_174x500_29shots_2000steps"
    nz, nx, h, dt, nt, freq = 174, 500, 10.0, 0.001, 2000, 8.0
    shots_per_gpu = 29
    pml = 20
    fwd_bytes, adj_bytes = 16.0, 20.0          # SURVEY.md 8d algorithmic B / cell-step

    def __init__(self, dev, rank, world, nt=None, shots=None, grid=None):
        import torch
        import physicsbasedfwi2_amd.compat.deepwave as deepwave
        from physicsbasedfwi2_amd import misfit
        self.torch, self.deepwave, self.dev, self.misfit = torch, deepwave, dev, misfit
        if nt:
            self.nt = nt
        if grid:
            self.nz, self.nx = grid
            self.name = "acoustic_%dx%d_%dshots_%dsteps" % (self.nz, self.nx, shots or self.shots_per_gpu, self.nt)
        ns = shots or self.shots_per_gpu
        self.ns = ns
        total = ns * world
        xs_all = np.linspace(0.0, (self.nx - 1) * self.h, total)
        xs = xs_all[rank * ns:(rank + 1) * ns]
        self.x_s = torch.zeros(ns, 1, 2)
        self.x_s[:, 0, 1] = torch.tensor(xs, dtype=torch.float32)
        self.x_r = torch.zeros(ns, self.nx, 2)
        self.x_r[:, :, 1] = (torch.arange(self.nx).float() * self.h)[None, :]
        self.x_s, self.x_r = self.x_s.to(dev), self.x_r.to(dev)
        self.wav = deepwave.wavelets.ricker(self.freq, self.nt, self.dt, 1.0 / self.freq) \
            .reshape(-1, 1, 1).repeat(1, ns, 1).to(dev)
        self.vp = torch.tensor(synth_vp(self.nz, self.nx, 0), device=dev, requires_grad=True)
        vp_true = torch.tensor(synth_vp(self.nz, self.nx, 1), device=dev)
        with torch.no_grad():
            obs = deepwave.scalar.Propagator({"vp": vp_true}, self.h, pml_width=self.pml)(
                self.wav, self.x_s, self.x_r, self.dt)
            omax, _ = obs.abs().max(dim=0, keepdim=True)
            self.obs = obs / (omax + 1e-10)
        self.t_fwd = self.t_bwd = 0.0
        self.n_timed = 0

    # cells one kernel launch updates (computational grid incl. absorbing layer, all shots)
    @property
    def cells_per_launch(self):
        return (self.nz + 2 * self.pml) * (self.nx + 2 * self.pml) * self.ns

    @property
    def units_per_step(self):
        return self.nz * self.nx * self.nt * self.ns

    def step(self, timed=False):
        torch = self.torch
        self.vp.grad = None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        prop = self.deepwave.scalar.Propagator({"vp": self.vp}, self.h, pml_width=self.pml)
        ev[0].record()
        rec = prop(self.wav, self.x_s, self.x_r, self.dt)
        ev[1].record()
        loss = self.misfit.l1_trace_normalized(rec, self.obs)     # fused HIP misfit + adjoint source
        (grec,) = torch.autograd.grad(loss, rec, retain_graph=True)
        ev[2].record()
        rec.backward(grec)
        ev[3].record()
        if timed:
            self._ev = getattr(self, "_ev", []) + [ev]
        return self.vp.grad, loss

    def kernel_family(self):
        from physicsbasedfwi2_amd.acoustic import AcousticPlan
        P = self.pml
        nw = AcousticPlan(self.nz + 2 * P, self.nx + 2 * P, self.nt, self.ns, 1, self.nx, 1, 1.0, 1.0,
                          self.dev.index or 0).cluster_slabs()
        return "single-launch time loop, %d row slabs per shot" % nw if nw else "one launch per step"

    def kernel_times(self):
        """avg per-launch duration (s) of the forward(+save) and adjoint(+imaging) kernels from
        the HIP events recorded on the launch stream around the two time loops."""
        tf = np.mean([e[0].elapsed_time(e[1]) for e in self._ev]) * 1e-3
        tb = np.mean([e[2].elapsed_time(e[3]) for e in self._ev]) * 1e-3
        return tf / self.nt, tb / max(1, self.nt - 1)

    def cpu_baseline(self, budget_s=15.0):
        """The C oracle (scalar port, OpenMP over shots) on a bounded sample of this workload:
        a short calibration run sizes the sample to about `budget_s` seconds of CPU work."""
        import oracle
        from oracle import helpers as H
        o = oracle.load("f32")
        cores = os.cpu_count() or 1
        P, h, dt = self.pml, self.h, self.dt
        vp = np.pad(synth_vp(self.nz, self.nx, 0), P, mode="edge").astype(np.float64)
        r = (vp * dt / h) ** 2
        N0, N1 = r.shape
        q0 = H.damp_profile_1d(N0, P, h) * h * h / (2 * dt)
        q1 = H.damp_profile_1d(N1, P, h) * h * h / (2 * dt)
        ns = min(self.ns, cores)
        sc, sw = H.cell_taps(np.full((ns, 1), P), (np.linspace(0, self.nx - 1, ns)[:, None]
                                                   .astype(int) + P), N1)
        rc, rw = H.cell_taps(np.full((ns, self.nx), P), np.arange(self.nx)[None, :]
                             .repeat(ns, 0) + P, N1)

        def run(nt):
            f = np.zeros((nt, ns, 1), dtype=np.float32)
            f[:, :, 0] = (H.ricker_deepwave(self.freq, nt, dt, 1.0 / self.freq) * h * h)[:, None]
            t0 = time.time()
            rec, G = o.acoustic_forward(r, q0, q1, f, sc, sw, rc, rw, save=True)
            o.acoustic_backward(r, q0, q1, sc, sw, rc, rw, rec, G)
            return time.time() - t0

        nt, reps, el = sized_cpu_sample(run, 100, self.nt, budget_s, bytes_per_step=4.0 * N0 * N1 * ns)
        return {"value": self.nz * self.nx * nt * ns * reps / el / 1e6, "unit": "Mcells*steps/s",
                "cores": min(cores, ns), "kind": "port",
                "sample": "%d passes of %d shots x %d steps of this workload, forward+adjoint, C oracle "
                          "(oracle/acoustic.c, OpenMP over shots), %.1f s" % (reps, ns, nt, el)}


def synth_elastic(nz, nx, seed, water_rows=26):
    """Vp as the acoustic case, Vs = Vp/sqrt(3) (0 in water), rho = 310 Vp^0.25
    (1000 in water)."""
    vp = synth_vp(nz, nx, seed, water_rows).astype(np.float64)
    vs = vp / np.sqrt(3.0)
    rho = 310.0 * vp ** 0.25
    vs[:water_rows] = 0.0
    rho[:water_rows] = 1000.0
    return vp.astype(np.float32), vs.astype(np.float32), rho.astype(np.float32)


class ElasticMarmousi:
    """BASELINE.json configs[2]: 2-D elastic Marmousi-II-like Vp/Vs/rho on the reference's
    100x300 grid at 20 m (networks.py:7314,7555), 32 shots, 3000 steps of 2 ms, 5 Hz source at
    40 m depth, 276 receivers at 460 m depth (networks.py:7612-7631), 10-node C-PML, L2 misfit on
    vx and vz, gradients w.r.t. Vp, Vs and rho."""
    name = "elastic_marmousi2_100x300_32shots_3000steps"
    nz, nx, h, dt, nt, freq = 100, 300, 20.0, 0.002, 3000, 5.0
    shots_per_gpu = 32
    pml = 10
    fwd_bytes, adj_bytes = 60.0, 80.0    # fused forward 60 B, adjoint + correlation 80 B
    # acquisition (networks.py:7612-7631): sources every 80 m at z = 40 m, receivers every 20 m at z = 460 m
    src_depth, rec_depth, rec_dx, x_first, x_margin, rec_x_max = 40.0, 460.0, 20.0, 380.0, 120.0, 5880.0
    free_surface = False

    def __init__(self, dev, rank, world, nt=None, shots=None, grid=None):
        import torch
        from physicsbasedfwi2_amd import elastic, misfit, profiles
        self.torch, self.elastic, self.dev, self.misfit = torch, elastic, dev, misfit
        if nt:
            self.nt = nt
        if os.environ.get("TUNE_PML"):
            self.pml = int(os.environ["TUNE_PML"])
        if grid:
            self.nz, self.nx = grid
            self.name = "elastic_%dx%d_%dshots_%dsteps" % (self.nz, self.nx, shots or self.shots_per_gpu, self.nt)
        ns = shots or self.shots_per_gpu
        self.ns = ns
        total = ns * world
        xs_all = np.linspace(self.x_first, (self.nx - 1) * self.h - self.x_margin, total)
        xs = xs_all[rank * ns:(rank + 1) * ns]
        _, _, sc = profiles.cells_round(xs, np.full(ns, self.src_depth), self.h, self.nx)
        xr = np.arange(self.x_first, min(self.rec_x_max, (self.nx - 2) * self.h) + self.h, self.rec_dx)
        _, _, rc = profiles.cells_round(xr, np.full(xr.size, self.rec_depth), self.h, self.nx)
        self.nrec = xr.size
        self.sc = torch.tensor(sc).view(ns, 1, 1)
        self.sw = torch.ones(ns, 1, 1)
        self.rc = torch.tensor(rc).view(1, -1, 1).repeat(ns, 1, 1)
        self.rw = torch.ones(ns, self.nrec, 1)
        wav = profiles.ricker(self.freq, self.nt, self.dt, 1.0 / self.freq) * (self.dt / self.h ** 2) * 1e9
        self.f = wav.reshape(-1, 1, 1).repeat(1, ns, 1).to(dev)
        vmax = 4500.0
        assert self.dt <= profiles.elastic_cfl_limit(self.h, vmax)
        self.pz = torch.tensor(profiles.cpml_tables(self.nz, self.pml, self.h, self.dt, 1500.0, 5.0,
                                                    low=not self.free_surface))
        self.px = torch.tensor(profiles.cpml_tables(self.nx, self.pml, self.h, self.dt, 1500.0, 5.0))
        self.prm = [torch.tensor(a, device=dev, requires_grad=True)
                    for a in synth_elastic(self.nz, self.nx, 0)]
        with torch.no_grad():
            true = [torch.tensor(a, device=dev) for a in synth_elastic(self.nz, self.nx, 1)]
            mat = elastic.staggered_materials(*true, self.dt, self.h, free_surface=self.free_surface)
            self.ox, self.oz = elastic.propagate(mat, self.f, self.pz, self.px, self.sc, self.sw,
                                                 self.rc, self.rw, self.pml, free_surface=self.free_surface)
        self._ev = []

    @property
    def cells_per_launch(self):
        return self.nz * self.nx * self.ns

    @property
    def units_per_step(self):
        return self.nz * self.nx * self.nt * self.ns

    def step(self, timed=False):
        torch = self.torch
        for p in self.prm:
            p.grad = None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        mat = self.elastic.staggered_materials(*self.prm, self.dt, self.h, free_surface=self.free_surface)
        ev[0].record()
        rvx, rvz = self.elastic.propagate(mat, self.f, self.pz, self.px, self.sc, self.sw, self.rc,
                                          self.rw, self.pml, free_surface=self.free_surface)
        ev[1].record()
        loss = self.misfit.l2_half(rvx, self.ox) + self.misfit.l2_half(rvz, self.oz)
        gx, gz = torch.autograd.grad(loss, [rvx, rvz], retain_graph=True)
        ev[2].record()
        torch.autograd.backward([rvx, rvz], [gx, gz])
        ev[3].record()
        if timed:
            self._ev.append(ev)
        return torch.stack([p.grad for p in self.prm]), loss

    def resident_nt(self):
        """None when all snapshots fit the default budget; otherwise a step count that does."""
        per_step = 20.0 * self.ns * self.nz * (4 * ((self.nx + 3) // 4))
        if self.nt * per_step <= self.elastic.DEFAULT_SNAPSHOT_BUDGET:
            return None
        return int(max(20, min(200, 0.25 * self.elastic.DEFAULT_SNAPSHOT_BUDGET // per_step)))

    def kernel_family(self):
        from physicsbasedfwi2_amd.elastic import ElasticPlan
        pl = ElasticPlan(self.nz, self.nx, self.nt, self.ns, 1, self.nrec, 1, self.pml, self.dev.index or 0,
                         0, int(self.free_surface))
        f, a = pl.cluster_slabs(False), pl.cluster_slabs(True)
        return "forward: %s; adjoint: %s" % tuple(
            "single-launch time loop, %d row slabs per shot" % n if n else "one launch per half step"
            for n in (f, a))

    def kernel_times(self):
        """avg duration (s) of one forward step (V+S launches) and one adjoint step (S^T+V^T)."""
        tf = np.mean([e[0].elapsed_time(e[1]) for e in self._ev]) * 1e-3
        tb = np.mean([e[2].elapsed_time(e[3]) for e in self._ev]) * 1e-3
        return tf / self.nt, tb / self.nt

    def cpu_baseline(self, budget_s=15.0):
        import oracle
        from oracle import helpers as H
        o = oracle.load("f32")
        cores = os.cpu_count() or 1
        vp, vs, rho = synth_elastic(self.nz, self.nx, 0)
        mat = H.elastic_materials(vp, vs, rho, self.dt, self.h, free_surface=self.free_surface)
        pz = H.cpml_profiles(self.nz, self.pml, self.h, self.dt, 1500.0, 5.0, lo=not self.free_surface)
        px = H.cpml_profiles(self.nx, self.pml, self.h, self.dt, 1500.0, 5.0)
        ns = min(self.ns, cores)
        sc = self.sc.numpy()[:ns]
        rc = self.rc.numpy()[:ns]

        def run(nt):
            f = np.zeros((nt, ns, 1), dtype=np.float32)
            f[:, :, 0] = (H.ricker_deepwave(self.freq, nt, self.dt, 1.0 / self.freq) * 1e6)[:, None]
            t0 = time.time()
            fs = int(self.free_surface)
            vx, vz, S = o.elastic_forward(mat, pz, px, f, sc, np.ones(sc.shape), rc, np.ones(rc.shape),
                                          save=True, free_surface=fs)
            o.elastic_backward(mat, pz, px, sc, np.ones(sc.shape), rc, np.ones(rc.shape), vx, vz, S,
                               free_surface=fs)
            return time.time() - t0

        nt, reps, el = sized_cpu_sample(run, 50, self.nt, budget_s, bytes_per_step=20.0 * self.nz * self.nx * ns)
        return {"value": self.nz * self.nx * nt * ns * reps / el / 1e6, "unit": "Mcells*steps/s",
                "cores": min(cores, ns), "kind": "port",
                "sample": "%d passes of %d shots x %d steps of this workload, forward+adjoint, C oracle "
                          "(oracle/elastic.c, OpenMP over shots), %.1f s" % (reps, ns, nt, el)}


class ElasticSEAM(ElasticMarmousi):
    """BASELINE.json configs[4] per-GPU share: 1000x3000 Vp/Vs/rho, 16 shots, 5000 steps
    (h = 30 m, dt = 2.5 ms, networks.py:9638,9810).  Snapshots do not fit: time checkpointing."""
    name = "elastic_seam_1000x3000_16shots_5000steps"
    nz, nx, h, dt, nt, freq = 1000, 3000, 30.0, 0.0025, 5000, 5.0
    shots_per_gpu = 16
    # free surface on, sources at z = 180 m, receivers every 30 m at z = 690 m (networks.py:9695-9723)
    src_depth, rec_depth, rec_dx, x_first, x_margin, rec_x_max = 180.0, 690.0, 30.0, 600.0, 600.0, 1e9
    free_surface = True


WORKLOADS = {"acoustic_marmousi": AcousticMarmousi, "elastic_marmousi": ElasticMarmousi,
             "elastic_seam": ElasticSEAM}


def measured_traffic(workload, kernel, cells):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_traffic.json, written by tools/pmc_traffic.py: 2 x FETCH_SIZE + WRITE_SIZE as
    MI355X_MICROARCH.md prescribes), normalised like `achieved` to one time step of all shots.
    None when no PMC summary exists for this workload."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as fh:
            rec = json.load(fh).get(workload, {}).get(kernel)
    except (OSError, ValueError):
        return None
    if not rec:
        return None
    return rec["bytes_per_cell_step"] * cells


def run_workload(name, args, dev, rank, world, want_cpu, grid=None, steps=None, warmup=None):
    import copy
    import torch
    import torch.distributed as dist
    kw = {}
    if grid or args.grid:
        kw["grid"] = grid or tuple(int(v) for v in args.grid.lower().split("x"))
    if steps is not None:                   # secondary workloads may time fewer passes (stated in their entry)
        args = copy.copy(args)
        args.steps, args.warmup = steps, warmup
    wl = WORKLOADS[name](dev, rank, world, nt=args.nt or None, shots=args.shots or None, **kw)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_step(timed):
        grad, loss = wl.step(timed)
        if world > 1:
            from physicsbasedfwi2_amd import dist as mdist
            mdist.all_reduce_gradient([grad], loss)
        return grad, loss

    for _ in range(args.warmup):
        one_step(False)
    barrier()
    losses, gsums = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        grad, loss = one_step(True)
        losses.append(loss.detach())            # device scalars: no sync inside the timed region
        gsums.append(grad.detach().double().abs().sum())
    barrier()
    el = time.perf_counter() - t0
    # same inputs every step: the gradient pass must reproduce itself bit for bit
    losses = [float(v) for v in losses]
    gsums = [float(v) for v in gsums]
    deterministic = all(v == losses[0] for v in losses) and all(v == gsums[0] for v in gsums)
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank != 0:
        return None
    value = wl.units_per_step * world * args.steps / el / 1e6
    t_f, t_b = wl.kernel_times()
    kernel_note = None
    nt_res = getattr(wl, "resident_nt", lambda: None)()
    if nt_res:
        # time-checkpointed run: the backward call re-runs the forward, so its events do not isolate the adjoint
        # kernels; the per-kernel durations come from a short run of the same workload with resident snapshots
        kw2 = dict(kw)
        short = WORKLOADS[name](dev, rank, world, nt=nt_res, shots=args.shots or None, **kw2)
        short.step(False)
        for _ in range(2):
            short.step(True)
        torch.cuda.synchronize()
        t_f, t_b = short.kernel_times()
        del short
        kernel_note = "per-kernel durations from a %d-step run of the same workload (snapshots resident)" % nt_res
    cells = wl.cells_per_launch
    interior = wl.nz * wl.nx * wl.ns          # the metric counts interior cells x user time steps
    kern = {
        "forward+save": {"avg_step_s": t_f, "alg_bytes_per_cell_step": wl.fwd_bytes,
                         "achieved_GBs": wl.fwd_bytes * cells / t_f / 1e9,
                         "Mcells_steps_per_s": interior / t_f / 1e6},
        "adjoint+imaging": {"avg_step_s": t_b, "alg_bytes_per_cell_step": wl.adj_bytes,
                            "achieved_GBs": wl.adj_bytes * cells / t_b / 1e9,
                            "Mcells_steps_per_s": interior / t_b / 1e6},
    }
    dom = "adjoint+imaging" if t_b >= t_f else "forward+save"
    traffic = measured_traffic(wl.name, dom, cells)
    out = {
        "metric": "grid-cells*timesteps/sec (forward+adjoint gradient pass)",
        "value": value, "unit": "Mcells*steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wl.name, "shots_per_gpu": wl.ns, "nt": wl.nt,
                   "grid": [wl.nz, wl.nx], "parallelism": "shots x%d" % world,
                   "kernel_family": wl.kernel_family()},
        "check": {"loss": losses[0], "grad_abs_sum": gsums[0], "bitwise_repeatable": deterministic},
        "roofline": {"bound": "hbm", "kernel": dom,
                     "achieved": kern[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": kern[dom]["achieved_GBs"] / HBM_PEAK_GBS,
                     "traffic": traffic},
        "kernels": kern,
    }
    if kernel_note:
        out["kernels_note"] = kernel_note
    try:                       # record the device and its clocks next to the numbers
        pr = torch.cuda.get_device_properties(dev)
        import ctypes
        from physicsbasedfwi2_amd import _lib
        clk = [ctypes.c_int32(0) for _ in range(3)]
        _lib.check(_lib.load().mifwi_device_info(dev.index or 0, *[ctypes.byref(c) for c in clk]))
        out["device"] = {"name": pr.name, "arch": getattr(pr, "gcnArchName", ""), "cus": clk[2].value,
                         "sclk_mhz": clk[0].value / 1e3, "mclk_mhz": clk[1].value / 1e3,
                         "hbm_gib": round(pr.total_memory / 2 ** 30, 1)}
    except Exception as exc:   # noqa: BLE001 - reporting only
        out["device"] = {"error": str(exc)}
    if want_cpu:
        out["cpu_baseline"] = wl.cpu_baseline()
    del wl
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--nt", type=int, default=0, help="override time steps (debug only)")
    ap.add_argument("--shots", type=int, default=0, help="override shots per GPU (debug only)")
    ap.add_argument("--grid", default="", help="NZxNX override (debug / large-grid measurements only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for --gpus > 1 (gloo: rehearsal of the multi-rank path)")
    ap.add_argument("--device-index", type=int, default=-1,
                    help="HIP device of this rank (default LOCAL_RANK; rehearsals put every rank on 0)")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the secondary (elastic) workload of the default invocation")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback in the product path)")
    if args.device_index >= 0:
        local = args.device_index
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    want_cpu = (not args.no_cpu_baseline) and world == 1
    primary = args.workload or "acoustic_marmousi"
    out = run_workload(primary, args, dev, rank, world, want_cpu)
    if args.workload is None and not args.no_also:
        # the north-star roofline target is stated on the elastic stencil: report it alongside
        keys = ("config", "value", "unit", "steps", "warmup", "ms_per_step", "check", "roofline", "kernels",
                "kernels_note", "cpu_baseline", "note")
        also = [run_workload("elastic_marmousi", args, dev, rank, world, want_cpu)]
        # BASELINE names no elastic grid - the same survey on the 10 m Marmousi-II grid 350x1700, where
        # the per-step kernels run and the 3000 snapshots do not fit (time checkpointing); fewer timed passes
        also.append(run_workload("elastic_marmousi", args, dev, rank, world, want_cpu, grid=(350, 1700),
                                 steps=min(args.steps, 3), warmup=1))
        if rank == 0:
            also[1]["note"] = "time-checkpointed: the forward runs twice per gradient pass (see kernels_note)"
            out["also"] = [{k: a[k] for k in keys if k in a} for a in also]
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
