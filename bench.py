#!/usr/bin/env python3
"""Benchmark of the FWI hot path (forward modelling -> misfit -> adjoint -> model gradient).

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A "step" is one full gradient evaluation over the rank's shots.  value = interior
cells x time steps x shots (all ranks) / wall second, in Mcells*steps/s (BASELINE.json metric).
Every rate in the line uses that one cell convention: INTERIOR cells of the physical grid (the
acoustic kernels also update the 20-cell sponge around it; those cells are work, not units).
Shots are independent => weak scaling: every rank runs the workload's per-GPU shot count and
the only collective is one all-reduce of the model gradient (RCCL) inside the timed step.
Inputs are synthetic and resident in HBM before timing starts.

`--gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself, as child processes,
before this process touches the GPU; under torchrun it is one of the ranks.

Default invocation: headline = elastic Marmousi-II (BASELINE configs[2], the stencil the
north-star roofline target is stated on); `also` = acoustic configs[1] and the elastic survey
on the 10 m grid 350x1700 (per-step kernels, HBM-bound).
"""
import argparse
import contextlib
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
PROFILE_ROUND = "r04"      # profiles/<round>_pmc_traffic.json, profiles/<round>_latency_floor.json


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


def csrc_sha16():
    """Fingerprint of the kernel sources: committed counter / ablation summaries are only quoted next to a
    bench line while they describe the kernels that produced it."""
    base = os.path.join(ROOT, "physicsbasedfwi2_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(base)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(base, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def _profile_record(fname, workload, kernel):
    """Record of (workload, kernel) in a committed profiles/ summary, or (None, reason)."""
    path = os.path.join(ROOT, "profiles", fname)
    try:
        with open(path) as fh:
            doc = json.load(fh)
    except (OSError, ValueError):
        return None, "no " + fname
    if doc.get("csrc_sha16") != csrc_sha16():
        return None, "%s describes other kernels (csrc %s, now %s)" % (fname, doc.get("csrc_sha16"), csrc_sha16())
    rec = doc.get(workload, {}).get(kernel)
    # the tag names the kernel sources the summary was measured on (their fingerprint, which is what was just compared) -
    # not a commit: the commit that holds a summary is by construction not the one it was measured at
    return (rec, "csrc " + doc["csrc_sha16"]) if rec else (None, "workload not in " + fname)


def measured_traffic(workload, kernel):
    """HBM bytes per interior cell-step of a time-loop kernel from the committed rocprofv3 PMC passes
    (profiles/<round>_pmc_traffic.json, written by tools/pmc_traffic.py: 2 x FETCH_SIZE + WRITE_SIZE as
    MI355X_MICROARCH.md prescribes).  (None, why) when no summary exists for the kernels as they are now."""
    rec, tag = _profile_record(PROFILE_ROUND + "_pmc_traffic.json", workload, kernel)
    return (rec["bytes_per_cell_step"], tag) if rec else (None, tag)


def latency_floor(workload, kernel):
    """Seconds per step of the single-launch kernel with hand-off waits and the snapshot stream ablated
    (profiles/<round>_latency_floor.json, tools/latency_floor.py: a -DMIFWI_ABLATIONS build): what is left is
    the chain of barrier-separated LDS phases, the bound of an LDS-resident time loop."""
    rec, tag = _profile_record(PROFILE_ROUND + "_latency_floor.json", workload, kernel)
    return (rec["floor_s_per_step"], tag) if rec else (None, tag)


def issue_bound(workload, kernel, t_step, sclk_hz=2.4e9):
    """The bound quoted for an LDS-resident time loop: issue cycles of the SHIPPED kernel over the cycles of a step.
    Instruction counts per wave and step come from the committed SQ counters (profiles/<round>_issue_counters.json,
    tools/issue_counters.py); a wave's vector, scalar or LDS instruction costs 4 issue cycles (MI355X_MICROARCH.md,
    "vector-instruction ISSUE cost").  frac_wave = how much of a step one wave spends issuing (the rest it waits: LDS
    and memory latency, barriers, hand-off polls); frac_simd_valu = how busy the SIMD's vector pipe is with the
    waves_per_simd waves that share it - 1.0 would be the vector-issue roofline of this instruction stream."""
    rec, tag = _profile_record(PROFILE_ROUND + "_issue_counters.json", workload, kernel)
    if not rec:
        return {"bound": "issue", "frac_wave": None, "note": tag}
    cyc = t_step * sclk_hz
    wps = 2 if workload.startswith("elastic") else 4             # 512 threads x 256 VGPRs / 1024 threads x 128 VGPRs
    per = rec["valu_per_wave_step"] + rec["salu_per_wave_step"] + rec["lds_per_wave_step"]
    return {"bound": "issue", "valu_per_wave_step": round(rec["valu_per_wave_step"], 1),
            "salu_per_wave_step": round(rec["salu_per_wave_step"], 1), "lds_per_wave_step": round(rec["lds_per_wave_step"], 1),
            "cycles_per_step": round(cyc), "issue_cycles_per_wave_step": round(4 * per),
            "frac_wave": 4 * per / cyc, "waves_per_simd": wps,
            "frac_simd_valu": wps * 4 * rec["valu_per_wave_step"] / cyc, "counters_profiled_at": tag,
            "note": "4 issue cycles per instruction; cycles at %.1f GHz" % (sclk_hz / 1e9)}


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


@contextlib.contextmanager
def _env(**kw):
    old = {k: os.environ.get(k) for k in kw}
    os.environ.update({k: str(v) for k, v in kw.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


class AcousticMarmousi:
    """BASELINE.json configs[1]: 2-D acoustic Marmousi-like 174x500, 29 shots, 2000 steps,
    deepwave-shaped call protocol + the L1 trace-normalised misfit of networks.py:5467-5476."""
    name = "acoustic_marmousi_174x500_29shots_2000steps"
    nz, nx, h, dt, nt, freq = 174, 500, 10.0, 0.001, 2000, 8.0
    shots_per_gpu = 29
    pml = 10                                   # the width `Propagator({'vp': m}, dx)` gets (networks.py:5408: none passed)
    fwd_bytes, adj_bytes = 16.0, 20.0          # SURVEY.md 8d algorithmic B / cell-step, one launch per step
    # an LDS-resident time loop only has to move the snapshot stream (G^n, 4 B/cell-step out, 4 back in)
    resident_fwd_bytes, resident_adj_bytes = 4.0, 4.0

    def __init__(self, dev, rank, world, nt=None, shots=None, grid=None, span=None, absorbing=None, pml=None):
        import torch
        import physicsbasedfwi2_amd.compat.deepwave as deepwave
        from physicsbasedfwi2_amd import misfit
        self.torch, self.deepwave, self.dev, self.misfit = torch, deepwave, dev, misfit
        # absorbing="cpml" (the shim's default): `pml_width` is what it is in deepwave, the width of a PML (second-order
        # C-PML); "sponge" (or BENCH_ABSORBING=sponge): the reference's in-tree damping layer, the shim's opt-out
        self.absorbing = absorbing or os.environ.get("BENCH_ABSORBING", "cpml")
        if pml:
            self.pml = int(pml)
        if os.environ.get("BENCH_PML_WIDTH"):
            self.pml = int(os.environ["BENCH_PML_WIDTH"])
        self.full_nt = type(self).nt
        if nt:
            self.nt = nt
        if grid:
            self.nz, self.nx = grid
        ns = shots or self.shots_per_gpu
        total, lo = ns * world, rank * ns
        if span:                               # strong scaling: this rank's block [lo, hi) of `total` shots
            total, lo, hi = span
            ns = hi - lo
        if grid or shots or nt or span:
            self.name = "acoustic_%dx%d_%dshots_%dsteps" % (self.nz, self.nx, total if span else ns, self.nt)
        self.name += "_%s%d" % (self.absorbing.replace("-", ""), self.pml)
        self.ns = ns
        xs_all = np.linspace(0.0, (self.nx - 1) * self.h, total)
        xs = xs_all[lo:lo + ns]
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
            obs = deepwave.scalar.Propagator({"vp": vp_true}, self.h, pml_width=self.pml, absorbing=self.absorbing)(
                self.wav, self.x_s, self.x_r, self.dt)
            omax, _ = obs.abs().max(dim=0, keepdim=True)
            self.obs = obs / (omax + 1e-10)
        self._ev = []
        self.last_rec = None

    @property
    def profile_key(self):
        return "acoustic_%dx%d" % (self.nz, self.nx) + ("" if self.absorbing == "sponge" else "_" + self.absorbing)   # keys of profiles/*.json

    @property
    def interior_cells(self):
        return self.nz * self.nx * self.ns

    @property
    def units_per_step(self):
        return self.nz * self.nx * self.nt * self.ns

    def step(self, timed=False):
        torch = self.torch
        self.vp.grad = None
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        prop = self.deepwave.scalar.Propagator({"vp": self.vp}, self.h, pml_width=self.pml, absorbing=self.absorbing)
        ev[0].record()
        rec = prop(self.wav, self.x_s, self.x_r, self.dt)
        ev[1].record()
        loss = self.misfit.l1_trace_normalized(rec, self.obs)     # fused HIP misfit + adjoint source
        (grec,) = torch.autograd.grad(loss, rec, retain_graph=True)
        ev[2].record()
        rec.backward(grec)
        ev[3].record()
        if timed:
            self._ev.append(ev)
        self.last_rec = rec.detach()
        return self.vp.grad, loss

    def _slabs(self):
        from physicsbasedfwi2_amd.acoustic import AcousticPlan
        P = self.pml
        pl = AcousticPlan(self.nz + 2 * P, self.nx + 2 * P, self.nt, self.ns, 1, self.nx, 1, 1.0, 1.0,
                          self.dev.index or 0, 0, P, P if self.absorbing == "cpml" else 0)
        nw = pl.cluster_slabs()
        pl.close()
        return nw, nw

    def resident(self):
        """(forward, adjoint): True where the time loop runs LDS-resident in one launch."""
        f, a = self._slabs()
        return bool(f), bool(a)

    def kernel_family(self):
        nw = self._slabs()[0]
        return "single-launch time loop, %d row slabs per shot" % nw if nw else "one launch per step"

    def other_family_env(self):
        """Environment that makes the plan pick the other kernel family (the in-run cross-check)."""
        return ({"MIFWI_AC_CLUSTER": "0"}, "one launch per step") if self._slabs()[0] else (None, None)

    def resident_nt(self):
        return None

    def kernel_times(self):
        """avg per-step duration (s) of the forward(+save) and adjoint(+imaging) time loops from the HIP
        events recorded on the launch stream around them."""
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

        cpml = self.absorbing == "cpml"
        if cpml:                                   # the layer the GPU path runs: oracle/acoustic_cpml.c, same tables
            vmax, fpml = 50.0 * np.ceil(float(vp.max()) / 50.0 - 1e-9), 0.25 / dt / 5.0      # scalar._pml_velocity
            ab0 = np.stack(H.cpml_profiles(N0, P, h, dt, vmax, fpml)[:2])
            ab1 = np.stack(H.cpml_profiles(N1, P, h, dt, vmax, fpml)[:2])

        def run(nt):
            f = np.zeros((nt, ns, 1), dtype=np.float32)
            f[:, :, 0] = (H.ricker_deepwave(self.freq, nt, dt, 1.0 / self.freq) * h * h)[:, None]
            t0 = time.time()
            if cpml:
                rec, G = o.acoustic_cpml_forward(r, ab0, ab1, f, sc, sw, rc, rw, save=True)
                o.acoustic_cpml_backward(r, ab0, ab1, sc, sw, rc, rw, rec, G)
            else:
                rec, G = o.acoustic_forward(r, q0, q1, f, sc, sw, rc, rw, save=True)
                o.acoustic_backward(r, q0, q1, sc, sw, rc, rw, rec, G)
            return time.time() - t0

        nt, reps, el = sized_cpu_sample(run, 100, self.nt, budget_s, bytes_per_step=4.0 * N0 * N1 * ns)
        return {"value": self.nz * self.nx * nt * ns * reps / el / 1e6, "unit": "Mcells*steps/s",
                "cores": min(cores, ns), "kind": "port",
                "sample": "%d passes of %d shots x %d steps of this workload, forward+adjoint, C oracle "
                          "(oracle/%s, OpenMP over shots), %.1f s" % (reps, ns, nt, "acoustic_cpml.c" if cpml else
                                                                      "acoustic.c", el)}


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
    resident_fwd_bytes, resident_adj_bytes = 20.0, 20.0      # LDS-resident: the five snapshot planes only
    # acquisition (networks.py:7612-7631): sources every 80 m at z = 40 m, receivers every 20 m at z = 460 m
    src_depth, rec_depth, rec_dx, x_first, x_margin, rec_x_max = 40.0, 460.0, 20.0, 380.0, 120.0, 5880.0
    free_surface = False

    def __init__(self, dev, rank, world, nt=None, shots=None, grid=None, span=None):
        import torch
        from physicsbasedfwi2_amd import elastic, misfit, profiles
        self.torch, self.elastic, self.dev, self.misfit = torch, elastic, dev, misfit
        self.full_nt = type(self).nt
        if nt:
            self.nt = nt
        if os.environ.get("TUNE_PML"):
            self.pml = int(os.environ["TUNE_PML"])
        if grid:
            self.nz, self.nx = grid
        ns = shots or self.shots_per_gpu
        total, lo = ns * world, rank * ns
        if span:                               # strong scaling: this rank's block [lo, hi) of `total` shots
            total, lo, hi = span
            ns = hi - lo
        if grid or shots or nt or span:
            self.name = "elastic_%dx%d_%dshots_%dsteps" % (self.nz, self.nx, total if span else ns, self.nt)
        self.ns = ns
        # A shortened time axis (kernel measurements on the big grids) would end before the source wavelet has
        # peaked and long before anything has come back from below the water layer, which the model and the
        # "observed" model share: misfit and gradient would be exact zeros.  Such samples compress the
        # acquisition - sources four rows below the sea bed, receivers three rows further down, wavelet peak
        # within the first sixth of the run - so that the timed kernels work on real numbers.
        t_arrive = 2.0 * (26 * self.h - min(self.src_depth, self.rec_depth)) / 1500.0 + 1.5 / self.freq
        self.sample_acquisition = self.nt * self.dt < 1.5 * t_arrive
        src_depth, rec_depth, freq = self.src_depth, self.rec_depth, self.freq
        if self.sample_acquisition:
            src_depth = 30 * self.h
            rec_depth = src_depth + 3 * self.h
            freq = max(self.freq, 6.0 / (self.nt * self.dt))
        xs_all = np.linspace(self.x_first, (self.nx - 1) * self.h - self.x_margin, total)
        xs = xs_all[lo:lo + ns]
        _, _, sc = profiles.cells_round(xs, np.full(ns, src_depth), self.h, self.nx)
        xr = np.arange(self.x_first, min(self.rec_x_max, (self.nx - 2) * self.h) + self.h, self.rec_dx)
        _, _, rc = profiles.cells_round(xr, np.full(xr.size, rec_depth), self.h, self.nx)
        self.nrec = xr.size
        # acquisition and C-PML profiles live on the device, as the model does (a training loop builds them once)
        self.sc = torch.tensor(sc, dtype=torch.int32).view(ns, 1, 1).to(dev)
        self.sw = torch.ones(ns, 1, 1, device=dev)
        self.rc = torch.tensor(rc, dtype=torch.int32).view(1, -1, 1).repeat(ns, 1, 1).to(dev)
        self.rw = torch.ones(ns, self.nrec, 1, device=dev)
        wav = profiles.ricker(freq, self.nt, self.dt, 1.0 / freq) * (self.dt / self.h ** 2) * 1e9
        self.f = wav.reshape(-1, 1, 1).repeat(1, ns, 1).to(dev)
        vmax = 4500.0
        assert self.dt <= profiles.elastic_cfl_limit(self.h, vmax)
        self.pz = torch.tensor(profiles.cpml_tables(self.nz, self.pml, self.h, self.dt, 1500.0, 5.0,
                                                    low=not self.free_surface), dtype=torch.float32).to(dev)
        self.px = torch.tensor(profiles.cpml_tables(self.nx, self.pml, self.h, self.dt, 1500.0, 5.0), dtype=torch.float32).to(dev)
        self.prm = [torch.tensor(a, device=dev, requires_grad=True)
                    for a in synth_elastic(self.nz, self.nx, 0)]
        with torch.no_grad():
            true = [torch.tensor(a, device=dev) for a in synth_elastic(self.nz, self.nx, 1)]
            mat = elastic.staggered_materials(*true, self.dt, self.h, free_surface=self.free_surface)
            self.ox, self.oz = elastic.propagate(mat, self.f, self.pz, self.px, self.sc, self.sw,
                                                 self.rc, self.rw, self.pml, free_surface=self.free_surface)
        self._ev = []
        self._ev_chunks = []
        self._last = None

    @property
    def last_rec(self):
        """[2][nt][shots][nrec]: vx and vz of the last pass (the in-run cross-check compares them between kernel families)."""
        torch = self.torch
        if self._last is None:
            return None
        if isinstance(self._last, tuple):
            return torch.stack(list(self._last))
        return torch.cat(self._last, dim=2)

    @property
    def profile_key(self):
        return "elastic_%dx%d%s" % (self.nz, self.nx, "_fs" if self.free_surface else "")

    @property
    def interior_cells(self):
        return self.nz * self.nx * self.ns

    @property
    def units_per_step(self):
        return self.nz * self.nx * self.nt * self.ns

    def shot_chunk(self):
        """0: all shots in one call (snapshots resident, or time-checkpointed if they do not fit); k > 0: the snapshots
        of the whole time axis do not fit for all shots but do for k at a time - the gradient pass then takes the shots
        k at a time, forward straight into adjoint, instead of paying a second forward sweep for time checkpoints
        (elastic.gradient_in_shot_chunks; the misfit is known, the observed data are in hand)."""
        if os.environ.get("MIFWI_BENCH_CHUNKS", "1") == "0":
            return 0
        if getattr(self, "_chunk", None) is not None:
            return self._chunk
        # the bench owns the device: most of the free memory may hold snapshots (a training loop that shares the GPU with
        # a CNN would pass its own budget); chunks of equal size, as few as fit
        k = self.elastic.resident_shot_chunk(self.ns, self.nt, self.nz, self.nx, snapshot_budget=1 << 60, device=self.dev)
        if os.environ.get("MIFWI_BENCH_CHUNK"):
            k = int(os.environ["MIFWI_BENCH_CHUNK"])
        # as few chunks as fit, of (nearly) equal size; they share one snapshot tensor (elastic.snapshot_arena)
        k = -(-self.ns // -(-self.ns // k)) if 1 <= k < self.ns else 0
        self._chunk = k if k >= 2 else 0
        return self._chunk

    def _step_chunked(self, chunk, timed):
        torch = self.torch
        mat = self.elastic.staggered_materials(*self.prm, self.dt, self.h, free_surface=self.free_surface)
        recs, t_f, t_b = [], [], []

        def loss_fn(rvx, rvz, sl):
            recs.append(torch.stack([rvx.detach(), rvz.detach()]))
            return (self.misfit.l2_half(rvx, self.ox[:, sl].contiguous()) +
                    self.misfit.l2_half(rvz, self.oz[:, sl].contiguous()))

        # per-chunk events around the forward and the adjoint time loops (the autograd hooks of propagate)
        leaf = mat.detach().requires_grad_(True)
        total = None
        # the chunks - of this pass and of the next ones - share the snapshot tensor of the first (largest) one: freed and
        # re-requested per pass, torch's allocator splits the cached 200 GB block for smaller requests in between
        if getattr(self, "_arena", None) is None:
            self._arena = self.elastic.snapshot_arena()
        self._arena.__enter__()
        for a in range(0, self.ns, chunk):
            sl = slice(a, min(a + chunk, self.ns))
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
            rvx, rvz = self.elastic.propagate(leaf, self.f[:, sl], self.pz, self.px, self.sc[sl], self.sw[sl], self.rc[sl],
                                              self.rw[sl], self.pml, free_surface=self.free_surface,
                                              snapshot_budget=1 << 60)
            ev[1].record()
            loss = loss_fn(rvx, rvz, sl)
            gx, gz = torch.autograd.grad(loss, [rvx, rvz], retain_graph=True)
            ev[2].record()
            torch.autograd.backward([rvx, rvz], [gx, gz])
            ev[3].record()
            t_f.append((ev[0], ev[1])); t_b.append((ev[2], ev[3]))
            total = loss.detach() if total is None else total + loss.detach()
        self.elastic.snapshot_arena.current = None      # leave the arena, keep its tensor
        mat.backward(leaf.grad)
        if timed:
            self._ev_chunks.append((t_f, t_b))
        self._last = recs                               # [vx | vz] of every chunk, concatenated on demand (cross-check only)
        return torch.stack([p.grad for p in self.prm]), total

    def step(self, timed=False):
        torch = self.torch
        for p in self.prm:
            p.grad = None
        chunk = self.shot_chunk()
        if chunk:
            return self._step_chunked(chunk, timed)
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
        self._last = (rvx.detach(), rvz.detach())       # stacked on demand (cross-check only): not a 200 MB copy per timed pass
        return torch.stack([p.grad for p in self.prm]), loss

    def resident_nt(self):
        """None when all snapshots fit the default budget; otherwise a step count that does."""
        per_step = self.elastic.snapshot_bytes_per_cell() * self.ns * self.nz * (4 * ((self.nx + 3) // 4))
        if self.nt * per_step <= self.elastic.DEFAULT_SNAPSHOT_BUDGET:
            return None
        return int(max(20, min(200, 0.25 * self.elastic.DEFAULT_SNAPSHOT_BUDGET // per_step)))

    def _plan_facts(self):
        """(slabs forward, slabs adjoint, kernel_flags) of the plan a call of this workload creates - for the chunk size
        when the shots are taken in chunks."""
        from physicsbasedfwi2_amd.elastic import ElasticPlan
        ns = self.shot_chunk() or self.ns
        pl = ElasticPlan(self.nz, self.nx, self.nt, ns, 1, self.nrec, 1, self.pml, self.dev.index or 0,
                         0, int(self.free_surface))
        out = pl.cluster_slabs(False), pl.cluster_slabs(True), int(pl.layout.kernel_flags)
        pl.close()
        return out

    def resident(self):
        f, a, _ = self._plan_facts()
        return bool(f), bool(a)

    def kernel_family(self):
        f, a, flags = self._plan_facts()
        names = self.elastic.kernel_family(flags)
        return "forward: %s; adjoint: %s" % tuple(
            "%s, %d row slabs per shot" % (nm, n) if n else nm for nm, n in zip(names, (f, a)))

    def other_family_env(self):
        f, a, flags = self._plan_facts()
        if f or a:
            return {"MIFWI_EL_CLUSTER": "0", "MIFWI_EL_CLUSTER_ADJ": "0"}, "one launch per half step"
        return self.elastic.other_per_step_env(flags)

    def this_family_env(self):
        """Environment that makes a ONE-shot run (the cross-check) use the formulation the timed run used: the default
        of the per-step forward depends on the size of a launch."""
        f, a, flags = self._plan_facts()
        if f or a:
            return {}
        return {"MIFWI_EL_FUSED": "1" if flags & 4 else "0"}

    def kernel_times(self):
        """avg duration (s) of one forward step and one adjoint step of the time loops."""
        if self._ev_chunks:      # shots taken a few at a time: a step of ALL shots = the sum over the chunks
            tf = np.mean([sum(a.elapsed_time(b) for a, b in tf_) for tf_, _ in self._ev_chunks]) * 1e-3
            tb = np.mean([sum(a.elapsed_time(b) for a, b in tb_) for _, tb_ in self._ev_chunks]) * 1e-3
            return tf / self.nt, tb / self.nt
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
        sc = self.sc.cpu().numpy()[:ns]
        rc = self.rc.cpu().numpy()[:ns]

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


def kernel_report(wl_key, label, t_step, interior, streaming_bytes, resident_bytes, is_resident):
    """Rates of one time-loop kernel, all on interior cells.  `alg` = the algorithmic bytes of the
    formulation that runs: SURVEY 8d's per-step streaming figure for one launch per (half) step; for an
    LDS-resident time loop only the snapshot stream has to cross HBM."""
    alg = resident_bytes if is_resident else streaming_bytes
    hbm, tag = measured_traffic(wl_key, label)
    k = {"avg_step_s": t_step, "lds_resident": bool(is_resident),
         "alg_bytes_per_cell_step": alg, "alg_GBs": alg * interior / t_step / 1e9,
         "streaming_bytes_per_cell_step": streaming_bytes,
         "streaming_equivalent_GBs": streaming_bytes * interior / t_step / 1e9,
         "Mcells_steps_per_s": interior / t_step / 1e6,
         "hbm_bytes_per_cell_step_measured": hbm,
         "hbm_GBs_measured": None if hbm is None else hbm * interior / t_step / 1e9,
         "traffic_profiled_at": tag if hbm is not None else None,
         "traffic_counter_note": "2 x FETCH_SIZE + WRITE_SIZE at the L2's fabric side: traffic served by the Infinity "
                                 "Cache counts as well, so on cache-resident passes this is an upper bound of the HBM bytes"}
    if hbm is None:
        k["traffic_note"] = tag
    if is_resident:
        # the bound that applies to an LDS-resident loop, measured on the shipped kernel: issue cycles / step cycles
        k["issue"] = issue_bound(wl_key, label, t_step)
        floor, ftag = latency_floor(wl_key, label)
        if floor is not None:
            k["latency"] = {"bound": "latency", "floor_us_per_step": floor * 1e6,
                            "achieved_us_per_step": t_step * 1e6, "frac": min(1.0, floor / t_step),
                            "floor_profiled_at": ftag,
                            "note": "diagnostic, not a roofline: the same kernel with hand-off waits and the snapshot "
                                    "stream ablated (what the waits cost); the bound is `issue`"}
        else:
            k["latency"] = {"bound": "latency", "floor_us_per_step": None, "note": ftag}
    return k


def roofline_of(kern, dom, interior):
    """The contract's roofline object for the dominant kernel: achieved = algorithmic bytes of one step of
    all shots / its duration (HIP events on the launch stream); traffic = measured HBM bytes of the same
    unit (committed PMC passes of these very kernels) or null."""
    k = kern[dom]
    r = {"bound": "hbm", "kernel": dom, "achieved": k["alg_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": k["alg_GBs"] / HBM_PEAK_GBS,
         "traffic": None if k["hbm_bytes_per_cell_step_measured"] is None
         else k["hbm_bytes_per_cell_step_measured"] * interior,
         "traffic_profiled_at": k["traffic_profiled_at"],
         "alg_bytes_per_launch_step": k["alg_bytes_per_cell_step"] * interior,
         "cells": "interior",
         "formulation": ("LDS-resident time loop in one launch: only the snapshot stream must cross HBM "
                         "(%.0f B/cell-step); the bound that applies is instruction issue, see `issue`"
                         % k["alg_bytes_per_cell_step"]) if k["lds_resident"] else
                        "one launch per (half) step: SURVEY 8d streaming bytes (%.0f B/cell-step)"
                        % k["alg_bytes_per_cell_step"]}
    if k["lds_resident"]:
        r["streaming_equivalent_GBs"] = k["streaming_equivalent_GBs"]
        r["issue"] = k.get("issue")
        r["latency"] = k.get("latency")
    assert r["frac"] <= 1.0 + 1e-9, "a roofline fraction above 1 means the byte count does not describe the kernel"
    return r


def cross_check(wl, name, dev, kw):
    """Untimed, after the timed loop: one shot of the same workload through the family that was timed and
    through the other one (environment switch read at plan creation); traces must agree bit for bit, the
    gradient to fp32 summation order."""
    import torch
    env, label = wl.other_family_env()
    if env is None:
        return {"verified": None, "note": "this plan has one kernel family only"}
    nt = min(wl.nt, 400) if wl.resident_nt() else wl.nt
    outs = []
    this = wl.this_family_env() if hasattr(wl, "this_family_env") else {}
    for e in (this, env):
        with _env(**e):
            one = WORKLOADS[name](dev, 0, 1, nt=nt, shots=1, **kw)
            grad, _ = one.step(False)
            outs.append((one.last_rec.clone(), grad.detach().clone()))
            del one
    torch.cuda.synchronize()
    (ra, ga), (rb, gb) = outs
    tr = float((ra - rb).abs().max())
    den = float(gb.double().norm())
    gr = float((ga.double() - gb.double()).norm()) / (den if den > 0 else 1.0)
    ok = bool(float(ra.abs().max()) > 0 and tr == 0.0 and gr <= 2e-5)
    return {"verified": ok, "against": label, "shots": 1, "nt": nt, "trace_max_abs_diff": tr,
            "gradient_rel_l2": gr}


def run_workload(name, args, dev, rank, world, want_cpu, grid=None, steps=None, warmup=None, nt=None, absorbing=None, pml=None):
    import copy
    import torch
    import torch.distributed as dist
    kw = {}
    if absorbing:
        kw["absorbing"] = absorbing
    if pml:
        kw["pml"] = pml
    if grid or args.grid:
        kw["grid"] = grid or tuple(int(v) for v in args.grid.lower().split("x"))
    if steps is not None:                   # secondary workloads may time fewer passes (stated in their entry)
        args = copy.copy(args)
        args.steps, args.warmup = steps, warmup
    if nt is not None:                      # ... or a sample of the time axis (kernel rates of a grid whose full run takes minutes)
        args = copy.copy(args)
        args.nt = nt
    strong = getattr(args, "scaling", "weak") == "strong"
    total_shots = 0
    if strong:
        # strong scaling: the configuration's own shot count (C2: 29, C3: 32; --total-shots overrides) split over the
        # ranks in balanced contiguous blocks - 29 shots on 8 ranks are 4,4,4,4,4,3,3,3
        from physicsbasedfwi2_amd import dist as mdist
        total_shots = args.total_shots or WORKLOADS[name].shots_per_gpu
        lo, hi = mdist.shot_partition_balanced(total_shots, rank, world)
        if hi <= lo:
            raise SystemExit("bench --scaling strong: rank %d of %d owns no shot of %d" % (rank, world, total_shots))
        kw["span"] = (total_shots, lo, hi)
    wl = WORKLOADS[name](dev, rank, world, nt=args.nt or None, shots=args.shots or None, **kw)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    rank_events = []

    def one_step(timed):
        # events on the launch stream: [start | gradient pass done | all-reduce done]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if timed else None
        if ev:
            ev[0].record()
        grad, loss = wl.step(timed)
        if ev:
            ev[1].record()
        if world > 1:
            from physicsbasedfwi2_amd import dist as mdist
            mdist.all_reduce_gradient([grad], loss)
        if ev:
            ev[2].record()
            rank_events.append(ev)
        return grad, loss

    for _ in range(args.warmup):
        one_step(False)
    barrier()
    torch.cuda.reset_peak_memory_stats(dev)
    losses, gsums = [], []
    from physicsbasedfwi2_amd import _lib as _mifwi_lib
    fallbacks0 = int(_mifwi_lib.load().mifwi_fallback_count())
    agent0 = int(_mifwi_lib.load().mifwi_agent_handoff_count())
    slow0 = int(_mifwi_lib.load().mifwi_slow_handoff_count())
    t0 = time.perf_counter()
    kept = []
    for _ in range(args.steps):
        grad, loss = one_step(True)
        kept.append((grad.detach(), loss.detach()))      # looked at after the timed region (repeatability check below)
    barrier()
    el = time.perf_counter() - t0
    for grad, loss in kept:
        losses.append(loss)
        gsums.append(grad.abs().sum(dtype=torch.float64))
    del kept
    # single-launch time loops that gave up inside the timed region and were re-run with one launch per step (rank 0's
    # count): must be 0 for `kernel_family` to describe what was timed
    fallbacks = int(_mifwi_lib.load().mifwi_fallback_count()) - fallbacks0
    # time loops repeated with hand-offs through the fabric (placement check failed), and launches in which a slab waited
    # long for a neighbour (a GPU shared with another process): both 0 on a healthy, exclusively owned GPU
    agent_tier = int(_mifwi_lib.load().mifwi_agent_handoff_count()) - agent0
    slow_handoffs = int(_mifwi_lib.load().mifwi_slow_handoff_count()) - slow0
    # same inputs every step: the gradient pass must reproduce itself bit for bit
    losses = [float(v) for v in losses]
    gsums = [float(v) for v in gsums]
    deterministic = all(v == losses[0] for v in losses) and all(v == gsums[0] for v in gsums)
    # per-rank breakdown (outside the timed region): gradient pass and all-reduce of every rank, so that a scaling
    # curve explains itself - the slowest rank's pass, the spread, and what the one collective per pass costs
    pass_ms = sum(e[0].elapsed_time(e[1]) for e in rank_events) / max(1, len(rank_events))
    ar_ms = sum(e[1].elapsed_time(e[2]) for e in rank_events) / max(1, len(rank_events))
    ranks = {"world_size_seen": world, "backend": "none", "shots_per_rank": [wl.ns],
             "pass_ms_per_rank": [pass_ms], "all_reduce_ms_per_rank": [ar_ms]}
    units_all = float(wl.units_per_step) * (1 if strong else world)
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        mine = torch.tensor([pass_ms, ar_ms, float(wl.ns), float(wl.units_per_step)], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        allr = torch.stack(allr).cpu().numpy()
        ranks = {"world_size_seen": dist.get_world_size(), "backend": dist.get_backend(),
                 "shots_per_rank": [int(v) for v in allr[:, 2]],
                 "pass_ms_per_rank": [float(v) for v in allr[:, 0]],
                 "all_reduce_ms_per_rank": [float(v) for v in allr[:, 1]]}
        units_all = float(allr[:, 3].sum())
    ranks["pass_ms_min"], ranks["pass_ms_max"] = min(ranks["pass_ms_per_rank"]), max(ranks["pass_ms_per_rank"])
    ranks["all_reduce_ms_max"] = max(ranks["all_reduce_ms_per_rank"])
    ranks["all_reduce_note"] = ("event time on the launch stream from the end of this rank's pass to the end of its "
                                "all-reduce: includes waiting for the slowest rank")
    if rank != 0:
        return None
    if not args.timing_only and not (np.isfinite(losses[0]) and losses[0] > 1e-25 and gsums[0] > 1e-25):
        raise SystemExit("bench: the timed pass produced loss %.3g, |grad| sum %.3g - nothing reached the receivers, "
                         "the kernels were timed on zeros" % (losses[0], gsums[0]))
    value = units_all * args.steps / el / 1e6
    t_f, t_b = wl.kernel_times()
    kernel_note = None
    nt_res = wl.resident_nt()
    chunk = wl.shot_chunk() if hasattr(wl, "shot_chunk") else 0
    if chunk:
        nt_res = None          # every chunk ran with resident snapshots: its events isolate forward and adjoint
        kernel_note = ("snapshots of all %d steps fit for %d shots at a time: the gradient pass takes the shots in chunks "
                       "of %d, forward straight into adjoint (no time checkpoints, no second forward); per-kernel "
                       "durations = sums over the chunks" % (wl.nt, chunk, chunk))
    if nt_res:
        # time-checkpointed run: the backward call re-runs the forward, so its events do not isolate the adjoint
        # kernels; the per-kernel durations come from a short run of the same workload with resident snapshots
        short = WORKLOADS[name](dev, rank, world, nt=nt_res, shots=args.shots or None, **kw)
        short.step(False)
        for _ in range(2):
            short.step(True)
        torch.cuda.synchronize()
        t_f, t_b = short.kernel_times()
        del short
        kernel_note = "per-kernel durations from a %d-step run of the same workload (snapshots resident)" % nt_res
    interior = wl.interior_cells              # the metric counts interior cells x user time steps
    res_f, res_a = wl.resident()
    kern = {
        "forward+save": kernel_report(wl.profile_key, "forward+save", t_f, interior, wl.fwd_bytes,
                                      wl.resident_fwd_bytes, res_f),
        "adjoint+imaging": kernel_report(wl.profile_key, "adjoint+imaging", t_b, interior, wl.adj_bytes,
                                         wl.resident_adj_bytes, res_a),
    }
    dom = "adjoint+imaging" if t_b >= t_f else "forward+save"
    check = {"loss": losses[0], "grad_abs_sum": gsums[0], "bitwise_repeatable": deterministic, "fallbacks": fallbacks,
             "agent_scope_relaunches": agent_tier, "slow_handoff_launches": slow_handoffs}
    if not (args.no_verify or args.timing_only):
        check.update(cross_check(wl, name, dev, kw))
    out = {
        "metric": "grid-cells*timesteps/sec (forward+adjoint gradient pass)",
        "value": value, "unit": "Mcells*steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": wl.name, "shots_per_gpu": wl.ns, "nt": wl.nt,
                   "grid": [wl.nz, wl.nx],
                   "parallelism": ("%d shots over %d ranks (balanced contiguous blocks)" % (total_shots, world)
                                   if strong else "shots x%d" % world),
                   "cells": "interior cells of the physical grid in value, kernels and roofline alike",
                   "kernel_family": wl.kernel_family(),
                   "snapshots": getattr(wl, "elastic", None) and wl.elastic.snapshot_mode() or "f32"},
        "check": check,
        "memory": {"peak_allocated_GiB": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                   "peak_reserved_GiB": round(torch.cuda.max_memory_reserved(dev) / 2 ** 30, 2),
                   "note": "torch allocator, timed passes only (snapshots or checkpoints + work buffers + data)"},
        "ranks": ranks,
        "roofline": roofline_of(kern, dom, interior),
        "kernels": kern,
    }
    if getattr(wl, "sample_acquisition", False):
        out["config"]["acquisition"] = ("sample: %d of %d steps, sources 4 rows below the sea bed, receivers 3 rows "
                                        "further down, wavelet compressed into the run" % (wl.nt, wl.full_nt))
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


def _r(v, n=4):
    """Round floats to n significant digits (compact line); everything else unchanged."""
    if isinstance(v, float):
        return float("%.*g" % (n, v))
    if isinstance(v, dict):
        return {k: _r(x, n) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_r(x, n) for x in v]
    return v


def compact_roofline(r):
    """The contract's roofline object without the prose: bound, achieved, peak, unit, frac, traffic + what they
    are counted on.  `issue` (LDS-resident loops) is a DIAGNOSTIC of the vector pipe, not a roofline."""
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_profiled_at",
            "alg_bytes_per_launch_step", "cells", "streaming_equivalent_GBs")
    c = {k: r[k] for k in keep if k in r}
    iss = r.get("issue") or {}
    if iss.get("frac_simd_valu") is not None:
        c["issue_diagnostic"] = {"valu_per_wave_step": iss["valu_per_wave_step"], "waves_per_simd": iss["waves_per_simd"],
                                 "frac_simd_valu": iss["frac_simd_valu"]}
    return c


def compact_entry(e, headline):
    """What the final stdout line carries of one workload: the contract fields for the headline, a one-row summary
    (value, per-step kernel times, roofline fraction and traffic) for the secondary workloads."""
    kern = {}
    for name, k in e["kernels"].items():
        kern[name] = {"us_per_step": k["avg_step_s"] * 1e6, "alg_B_per_cell": k["alg_bytes_per_cell_step"],
                      "frac_of_hbm_peak": k["alg_GBs"] / HBM_PEAK_GBS,
                      "hbm_B_per_cell_measured": k["hbm_bytes_per_cell_step_measured"]}
    chk = e["check"]
    if headline:
        c = {k: e[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                               "scaling", "vs_baseline", "dtype", "data")}
        c["config"] = {k: e["config"][k] for k in ("workload", "shots_per_gpu", "nt", "grid", "parallelism",
                                                    "kernel_family", "snapshots") if k in e["config"]}
        c["config"]["cells"] = "interior"
        c["config"]["kernel_family"] = c["config"]["kernel_family"].replace("single-launch time loop", "single launch") \
            .replace(" row slabs per shot", " slabs/shot")
        c["check"] = {k: chk[k] for k in ("loss", "grad_abs_sum", "bitwise_repeatable", "fallbacks",
                                          "agent_scope_relaunches", "slow_handoff_launches", "verified",
                                          "trace_max_abs_diff", "gradient_rel_l2") if k in chk}
        rk = e["ranks"]
        c["ranks"] = {k: rk[k] for k in ("world_size_seen", "backend", "shots_per_rank", "pass_ms_per_rank",
                                         "all_reduce_ms_per_rank")}
        c["memory_peak_GiB"] = e["memory"]["peak_allocated_GiB"]
        c["roofline"] = compact_roofline(e["roofline"])
        c["kernels"] = kern
        if "cpu_baseline" in e:
            c["cpu_baseline"] = e["cpu_baseline"]
        if "device" in e:
            c["device"] = e["device"]
        return c
    c = {"workload": e["config"]["workload"], "grid": e["config"]["grid"], "shots": e["config"]["shots_per_gpu"],
         "nt": e["config"]["nt"], "steps": e["steps"], "value": e["value"], "ms_per_step": e["ms_per_step"],
         "kernels": kern, "roofline": {k: e["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic")},
         "verified": chk.get("verified"), "bitwise_repeatable": chk["bitwise_repeatable"], "fallbacks": chk["fallbacks"]}
    if "cpu_baseline" in e:
        c["cpu_Mcells_steps_per_s"] = e["cpu_baseline"]["value"]
    return c


MAX_LINE_BYTES = 6000        # the driver's parser lost round 3's 23 KB line: tests/test_bench_contract.py holds this bound


def final_line(out, also):
    """The ONE JSON line of the contract: the headline object (contract fields + roofline + cpu_baseline) and a one-row
    summary per secondary workload; everything else goes to the detail file."""
    line = compact_entry(out, True)
    if also:
        line["also"] = [compact_entry(a, False) for a in also]
    text = json.dumps(_r(line), separators=(",", ":"))
    if len(text) > MAX_LINE_BYTES:                   # never again a line the driver cannot take: drop the extras first
        line.pop("also", None)
        line.pop("device", None)
        text = json.dumps(_r(line), separators=(",", ":"))
    return text


def emit(out, also, detail_path=None):
    """Full report -> detail file (per-kernel issue / latency diagnostics, notes, every secondary workload in full);
    stdout gets exactly one compact JSON line, LAST."""
    keys = ("config", "value", "unit", "steps", "warmup", "ms_per_step", "check", "memory", "ranks", "roofline", "kernels",
            "kernels_note", "cpu_baseline", "note")
    full = dict(out)
    if also:
        full["also"] = [{k: a[k] for k in keys if k in a} for a in also]
    path = detail_path or os.environ.get("MIFWI_BENCH_DETAIL")
    if not path:
        scratch = os.path.join(ROOT, "gpurun_out")
        path = os.path.join(scratch if os.path.isdir(scratch) else ROOT, "bench_detail.json")
    try:
        with open(path, "w") as fh:
            json.dump(full, fh)
        print("bench.py: full report (per-kernel diagnostics, secondary workloads) -> %s" % path, file=sys.stderr)
    except OSError as exc:
        print("bench.py: detail file not written (%s)" % exc, file=sys.stderr)
    sys.stderr.flush()
    print(final_line(out, also))
    sys.stdout.flush()


def visible_gpus():
    """GPU agents of the KFD topology (nodes with SIMDs), capped by HIP_/ROCR_VISIBLE_DEVICES - counted from /sys so
    that the launching parent stays free of any HIP context; None when the topology cannot be read (the ranks then
    find out by themselves and the parent reports their exit codes)."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except (OSError, ValueError):
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def self_launch(args, argv):
    """--gpus N without a launcher: start N ranks of this script (one per GPU, rendezvous on 127.0.0.1) from a
    parent that never touches the GPU, pass rank 0's JSON line through, fail if any rank fails."""
    n = args.gpus
    if args.device_index < 0 and args.backend == "nccl":
        have = visible_gpus()                        # from /sys: the parent never loads the HIP runtime
        if have is not None and have < n:
            raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible" % (n, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        for r, p in enumerate(procs):
            code = p.wait()
            if code != 0 and rc == 0:
                rc = code
                print("bench.py: rank %d of %d exited with code %d" % (r, n, code), file=sys.stderr)
                for q in procs:                     # the others would wait for it at the next collective
                    if q.poll() is None:
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--nt", type=int, default=0, help="override time steps (debug only)")
    ap.add_argument("--shots", type=int, default=0, help="override shots per GPU (debug only)")
    ap.add_argument("--grid", default="", help="NZxNX override (debug / large-grid measurements only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for --gpus > 1 (gloo: rehearsal of the multi-rank path)")
    ap.add_argument("--device-index", type=int, default=-1,
                    help="HIP device of this rank (default LOCAL_RANK; rehearsals put every rank on 0)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the untimed cross-check of the two kernel families (counter / ablation runs)")
    ap.add_argument("--timing-only", action="store_true",
                    help="ablation builds (wrong results by construction): no cross-check, no refusal of a zero loss")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, the driver's contract): every rank runs the configuration's per-GPU shot count; "
                         "strong: the configuration's shots (--total-shots) are split over the ranks")
    ap.add_argument("--total-shots", type=int, default=0, help="shots of the whole job with --scaling strong")
    ap.add_argument("--detail", default="", help="where the full report goes (default gpurun_out/bench_detail.json)")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the secondary workloads of the default invocation")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))    # before anything here touches the GPU

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
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
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))

    want_cpu = (not args.no_cpu_baseline) and world == 1
    primary = args.workload or "elastic_marmousi"
    out = run_workload(primary, args, dev, rank, world, want_cpu)
    also = []
    if args.workload is None and not args.no_also:
        # C2 as the reference calls it: Propagator({'vp': m}, dx) - a PML of deepwave's default width (second-order C-PML,
        # 10 cells per side)
        also.append(run_workload("acoustic_marmousi", args, dev, rank, world, want_cpu, absorbing="cpml"))
        # ... and with the in-tree sponge of seisgan's model.py:6-29, 20 cells (the opt-out of the deepwave-shaped shim;
        # rounds 1-3 quoted C2 with it)
        also.append(run_workload("acoustic_marmousi", args, dev, rank, world, False, steps=min(args.steps, 5), warmup=1,
                                 absorbing="sponge", pml=20))
        # BASELINE names no elastic grid - the same survey on the 10 m Marmousi-II grid 350x1700, where
        # the per-step kernels run HBM-bound and the 3000 snapshots do not fit (time checkpointing); fewer passes
        also.append(run_workload("elastic_marmousi", args, dev, rank, world, want_cpu, grid=(350, 1700),
                                 steps=min(args.steps, 3), warmup=1))
        # BASELINE config 5's per-GPU share (1000x3000, free surface, 16 shots): a 90-step sample with resident
        # snapshots - the kernel rates of the SEAM-sized grid; the full 5000 steps take ~10 s per gradient pass
        # (time-checkpointed; profiles/<round>_c5_full_length.json)
        also.append(run_workload("elastic_seam", args, dev, rank, world, False, steps=min(args.steps, 3), warmup=1, nt=90))
        if rank == 0:
            also[1]["note"] = "C2 with the reference's in-tree sponge, 20 cells, instead of a PML (Propagator(..., pml_width=20, absorbing='sponge'))"
            also[2]["note"] = "snapshots of all shots do not fit at full length: see kernels_note for how the pass is cut"
            also[3]["note"] = ("90-step sample of the 5000-step configuration (snapshots resident): kernel rates of the "
                               "1000x3000 grid; the full-length pass is in profiles/%s_c5_full_length.json" % PROFILE_ROUND)
    if rank == 0:
        emit(out, also, args.detail)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
