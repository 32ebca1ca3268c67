/*
 * mifwi.h -- C-ABI of the MI355X-native 2-D finite-difference wave propagator
 * (libmifwi.so, hand-written HIP for gfx950).
 *
 * The reference (ADharaUTEXAS123007/PhysicsBasedFWI2) has no C interface of its own: its
 * hot path is reached through three third-party Python call protocols.  Every entry point
 * below names the reference call site(s) it replaces (paths relative to the reference
 * tree); INTEGRATION.md shows the binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success, a negative MIFWI_E* code otherwise; no C++
 *     exception crosses the boundary; mifwi_last_error() gives the message (thread-local).
 *   - all array arguments are DEVICE pointers owned by the caller (the Python host passes
 *     torch tensors' data_ptr()); the library allocates no device memory and keeps no
 *     global state; `stream` is a hipStream_t (0 = default stream).
 *   - arithmetic is fp32; indices are int32; cells are linear indices i0*n1+i1 into the
 *     computational grid (axis 1 fastest), -1 = inactive tap.
 *   - there is NO CPU fallback: without a visible gfx950 device every compute call fails
 *     with MIFWI_ENODEVICE.
 */
#ifndef MIFWI_H
#define MIFWI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIFWI_VERSION_MAJOR 0
#define MIFWI_VERSION_MINOR 7   /* 3: elastic desc gained snapshot_format, fd_order; layout snap_step_elems, snapshot_format; 4: mifwi_fallback_count; 5: mifwi_agent_handoff_count, mifwi_slow_handoff_count; acoustic desc gained cpml_width, layout state_elems; 6: mifwi_elastic_materials(_vjp), mifwi_acoustic_coefficients(_vjp); 7: mifwi_elastic_gradient_parametrization; elastic snapshot planes column-blocked in plans without a single-launch kernel */

enum {
    MIFWI_OK = 0,
    MIFWI_EINVAL = -1,     /* bad argument / shape mismatch                     */
    MIFWI_ENODEVICE = -2,  /* no HIP device / wrong architecture                */
    MIFWI_EHIP = -3,       /* a HIP runtime call failed                          */
    MIFWI_ECFL = -4,       /* stability limit violated                           */
    MIFWI_ENOMEM = -5      /* caller-provided workspace too small                */
};

const char *mifwi_last_error(void);
int mifwi_version(void);                 /* major*1000 + minor                  */
/* Calls of this process whose single-launch time loop gave up (hand-off time-out, slabs of a shot not on one XCD) and
 * were re-run with one launch per step: results are the same, the time is not.  Monitoring and tests read it; the
 * first such call is also noted on stderr (MIFWI_QUIET=1 silences that). */
int64_t mifwi_fallback_count(void);
/* Single-launch time loops that were repeated with hand-offs through the fabric (agent-scope publishes) because the
 * slabs of a shot had not been placed on one XCD: still one launch per time loop, 10-15 % slower per step. */
int64_t mifwi_agent_handoff_count(void);
/* Single-launch time loops in which some workgroup needed more than 32 poll passes for a neighbour's rows - what a
 * GPU shared with another process looks like (the loop then runs many times slower, results unchanged, no fall-back):
 * one rank must own the GPU while a propagator call runs.  The first such launch is noted on stderr. */
int64_t mifwi_slow_handoff_count(void);
int mifwi_device_count(void);            /* number of visible HIP devices, >=0  */
/* engine / memory clock (kHz) and compute units of a device, for measurement reports; 0 where unknown */
int mifwi_device_info(int device, int32_t *sclk_khz, int32_t *mclk_khz, int32_t *compute_units);

/* ======================================================================================
 * 2-D constant-density ACOUSTIC propagator (forward + exact discrete adjoint)
 *
 * Replaces:
 *   deepwave.scalar.Propagator({'vp': m}, dx)(src, x_s, x_r, dt) and its autograd
 *   backward                                   models/networks.py:5408-5411, 5449, 5464, 5491
 *   Devito Forward/Gradient operators          seisgan/fwi/pde/seismic/acoustic/operators.py:54-89,
 *                                              127-165 ; wavesolver.py:72-107, 143-172
 *
 * Scheme:  r = vp^2 dt^2/h^2 = s^2/(m h^2);  q = q0[i0]+q1[i1]
 *   u+ = (2u - (1-q r) u- + r L4(u)) / (1+q r) + sum_taps w f[n] r[cell]
 *   rec[n] = sum_taps w u^n[cell]            (time loop n = 0..nt-1)
 * ==================================================================================== */
typedef struct {
    int32_t n0, n1;        /* computational grid (caller has already padded the model)      */
    int32_t nt;            /* time steps                                                    */
    int32_t nshot;         /* shots propagated together (independent wavefields)            */
    int32_t nsrc, nrec;    /* sources / receivers per shot                                  */
    int32_t ntap;          /* taps per point: 1 = nearest cell, 4 = bilinear                */
    float c0, c1;          /* (h/h0)^2, (h/h1)^2 with h = min(h0,h1)                        */
    int32_t shots_per_group; /* shots marched by one thread (gradient RMW amortisation); 0 = auto */
    int32_t edge_rows;     /* optional hint: rows of absorbing layer at the top and at the bottom
                              (0 = unknown); lets the single-launch kernels give the layer slabs
                              of its own.  Results do not depend on it.                       */
    int32_t cpml_width;    /* 0: absorbing layer = the sponge q0, q1.  W > 0: second-order convolutional
                              PML, W cells wide on all four sides (what deepwave.scalar.Propagator's
                              pml_width is, models/networks.py:5408-5411):
                                psi_d <- b psi_d + a d_d u;  zeta_d <- b zeta_d + a (d_d^2 u + d_d psi_d)
                                u_tt = vp^2 sum_d [d_d^2 u + d_d psi_d + zeta_d]
                              with its exact transposed adjoint (DESIGN.md section 3,
                              oracle/acoustic_cpml.c).  The q0 / q1 arguments of the calls then carry the
                              layer's profiles instead of the sponge: q0 = [2][n0] (a, b along axis 0),
                              q1 = [2][gp] (a, b along axis 1), zero outside the layer.  One launch per
                              step family; the wavefield state grows by the memory variables
                              (layout.state_elems).                                              */
} mifwi_acoustic_desc;

typedef struct mifwi_acoustic_plan mifwi_acoustic_plan;

int mifwi_acoustic_plan_create(mifwi_acoustic_plan **plan, int device,
                               const mifwi_acoustic_desc *desc);
int mifwi_acoustic_plan_destroy(mifwi_acoustic_plan *plan);

/* Buffer layout the caller needs for allocation (all counts in floats):
 *   wavefields      [nshot][n0+4][pitch]   (2 halo rows top/bottom, interior column 0 at +4)
 *   coefficient r, every snapshot slice, gradients, accumulators   [n0][gp]
 *   gp = n1 rounded up to a multiple of 4; cells i1 >= n1 of r must be 0.                   */
typedef struct {
    int32_t gp, pitch, ngroups, shots_per_group;
    int64_t field_elems;          /* one wavefield set                                       */
    int64_t coef_elems;           /* n0*gp                                                   */
    int64_t work_forward_elems;   /* size of `work` for mifwi_acoustic_forward               */
    int64_t work_backward_elems;  /* size of `work` for mifwi_acoustic_backward              */
    int64_t state_elems;          /* head of `work` that is the forward state a checkpoint must keep:
                                     2*field_elems (+ the C-PML memory variables)             */
} mifwi_acoustic_layout;

int mifwi_acoustic_plan_layout(const mifwi_acoustic_plan *plan, mifwi_acoustic_layout *out);
/* Shot groups per pass of the per-step kernels (forward, adjoint); introspection only. */
int mifwi_acoustic_plan_pass_sizes(const mifwi_acoustic_plan *plan, int32_t *forward_groups, int32_t *adjoint_groups);

/* flags for the time-range calls */
#define MIFWI_ZERO_STATE 1   /* zero the wavefield state (and accumulators) in `work` first   */
#define MIFWI_FINALIZE   2   /* backward: reduce accumulators into grad_r, emit grad_f[k_lo-1] */

/* Forward modelling of steps n = n_begin .. n_end-1 (a full run is [0, nt) with
 * MIFWI_ZERO_STATE).  The wavefield state (u^n, u^{n-1}) lives in the first
 * 2*field_elems floats of `work`; buffer parity is absolute in n, so a run can be split
 * into ranges and the state copied out/in by the caller (time checkpointing).
 *   r [n0][gp], q0 [n0], q1 [gp]
 *   f [nt][nshot][nsrc]; src_cell/src_w [nshot][nsrc][ntap]; rec_cell/rec_w [nshot][nrec][ntap]
 *   rec_out [nt][nshot][nrec] or NULL (no sampling)
 *   snap : NULL, or [n_end-n_begin][nshot][n0][gp] receiving G^n = d u^{n+1}/d r
 *   work : layout.work_forward_elems floats                                                 */
int mifwi_acoustic_forward(mifwi_acoustic_plan *plan, const float *r, const float *q0,
                           const float *q1, const float *f, const int32_t *src_cell,
                           const float *src_w, const int32_t *rec_cell, const float *rec_w,
                           float *rec_out, float *snap, float *work, int32_t n_begin,
                           int32_t n_end, int32_t flags, void *stream);

/* Born / linearised modelling (seisgan/fwi/pde/seismic/acoustic/operators.py:168-207,
 * wavesolver.py:174-209): first-order change of the seismograms, drec = J dr, around the forward
 * run whose snapshots are passed in.  The perturbation field obeys the forward recursion with the
 * distributed source G^n dr and no point source; J is the exact transpose partner of the gradient
 * mifwi_acoustic_backward returns (<J dr, g> = <dr, grad_r(g)>).
 *   dr [n0][gp] perturbation of r;  snap as written by mifwi_acoustic_forward (step n at
 *   snap + (n-snap_first)*nshot*n0*gp);  drec_out [nt][nshot][nrec];  work = work_forward_elems   */
int mifwi_acoustic_born(mifwi_acoustic_plan *plan, const float *r, const float *q0, const float *q1,
                        const float *dr, const int32_t *rec_cell, const float *rec_w, const float *snap,
                        int32_t snap_first, float *drec_out, float *work, int32_t n_begin, int32_t n_end,
                        int32_t flags, void *stream);

/* Exact discrete adjoint + imaging for k = k_hi down to k_lo (a full run is k_hi = nt-1,
 * k_lo = 1 with MIFWI_ZERO_STATE | MIFWI_FINALIZE).  Step k needs snapshot G^{k-1}, found at
 * snap + (k-1-snap_first)*nshot*n0*gp.
 *   grad_rec [nt][nshot][nrec] = dJ/d rec
 *   grad_r [n0][gp]  (written when MIFWI_FINALIZE; sum over the plan's shots)
 *   grad_f NULL or [nt][nshot][nsrc]
 *   work : layout.work_backward_elems floats (adjoint state + accumulators)                 */
int mifwi_acoustic_backward(mifwi_acoustic_plan *plan, const float *r, const float *q0,
                            const float *q1, const int32_t *src_cell, const float *src_w,
                            const int32_t *rec_cell, const float *rec_w, const float *grad_rec,
                            const float *snap, int32_t snap_first, float *grad_r, float *grad_f,
                            float *work, int32_t k_hi, int32_t k_lo, int32_t flags,
                            void *stream);

/* ======================================================================================
 * 2-D P-SV ELASTIC velocity-stress propagator (forward + exact discrete adjoint)
 *
 * Replaces the DENISE-Black-Edition runs behind pyapi_denise:
 *   d.forward(model, src, rec) / d.grad(model, src, rec)   models/networks.py:7787, 9853-9877
 *   (parameter block 7698-7731 / 9790-9833; gradient read-back 7799-7802)
 *
 * Scheme: standard staggered grid, 4th-order space, leapfrog time,
 * C-PML memory variables on every derivative (layer inside the grid, pml_width nodes),
 * explosive source into sxx/szz, receivers sample vx/vz after the velocity update.
 *   mat [5][nz][gp] = lambda dt/h, (lambda+2mu) dt/h, mu_xz dt/h, dt/(h rho_x), dt/(h rho_z)
 *                     (staggered averages are formed by the host; columns >= nx must be 0)
 *   pz [6][nz], px [6][gp] : C-PML a, b, 1/kappa at integer then at half nodes
 *                     (a = b = 0, 1/kappa = 1 outside the layer and in columns >= nx)
 * ==================================================================================== */
typedef struct {
    int32_t nz, nx;          /* grid, z = depth is the slow axis                              */
    int32_t nt, nshot, nsrc, nrec, ntap;
    int32_t pml_width;       /* C-PML nodes per side (0 = none); profiles may still be zero
                                on a side (e.g. free surface)                                 */
    int32_t free_surface;    /* 1: stress-imaging free surface on row 0 (DENISE FREE_SURF): szz = 0
                                there, stresses mirrored oddly above it; the caller passes row 0 of
                                mat[0..1] in the effective form (0, M - L^2/M)                    */
    int32_t shots_per_group; /* adjoint: shots sharing one gradient accumulator; 0 = auto     */
    int32_t source_type;     /* 0: explosive, f added to sxx and szz after S (DENISE QUELLTYPB 1);
                                1 / 2: point force, f added to vx / vz between V and S (QUELLTYPB 2 / 3,
                                pyapi_denise attribute at networks.py:10419-10453).  f arrives scaled by the
                                host in every case; grad_f is the matching adjoint sample.  Force sources
                                run on the one-launch-per-half-step kernels.                          */
    int32_t record_pressure; /* 1: the plan also serves pressure receivers (DENISE SEISMO 2 / 4, adjoint source
                                type QUELLTYPB 4) through mifwi_elastic_plan_bind_pressure; such plans run on
                                the one-launch-per-half-step kernels.                                     */
    int32_t snapshot_format; /* MIFWI_SNAPSHOT_F32 (exact discrete adjoint) or MIFWI_SNAPSHOT_BF16: the five
                                forward snapshot planes are rounded to bf16 on their way to memory (10 instead of
                                20 B per cell-step each way, twice the steps per checkpoint segment; material
                                gradients within 4e-3 rel-L2 of the f32 form in the worst case, seismograms unchanged) - the
                                counterpart of the wavefield compression / decimation DENISE and deepwave apply to
                                their stored fields.  Honoured by plans that run the per-step
                                kernels; single-launch plans keep f32 (layout.snapshot_format says which).     */
    int32_t fd_order;        /* spatial order of the staggered-grid first derivatives (DENISE FD_ORDER, left commented at
                                models/networks.py:10447): 4 (Taylor weights 9/8, -1/24; 0 means 4) or 2 (1, 0).
                                Higher orders need a wider halo than the state layout carries: EINVAL.             */
} mifwi_elastic_desc;

enum { MIFWI_SNAPSHOT_F32 = 0, MIFWI_SNAPSHOT_BF16 = 1 };

typedef struct {
    int32_t gp, pitch, ngroups, shots_per_group;
    int64_t coef_elems;           /* nz*gp: one material / snapshot / gradient plane           */
    int64_t state_elems;          /* forward state (5 fields + memory variables) at the start
                                     of `work`: what a time checkpoint copies                  */
    int64_t work_forward_elems;
    int64_t work_backward_elems;
    int64_t snap_step_elems;      /* floats one time step of the snapshot buffer takes (all shots)  */
    int32_t snapshot_format;      /* the format this plan writes and reads                          */
    int32_t kernel_flags;         /* MIFWI_EL_KERNEL_*: which formulation of the time loops the plan picked   */
} mifwi_elastic_layout;

/* kernel_flags: forward / adjoint as ONE launch for the whole time loop (LDS-resident slabs), or - on grids that do not
   fit - V+S (S^T+V^T) fused into one launch per step; neither bit = one launch per half step */
enum { MIFWI_EL_KERNEL_FWD_SINGLE_LAUNCH = 1, MIFWI_EL_KERNEL_ADJ_SINGLE_LAUNCH = 2, MIFWI_EL_KERNEL_FWD_FUSED_STEP = 4,
       MIFWI_EL_KERNEL_ADJ_FUSED_STEP = 8,
       /* the single-launch loop takes the outer cells of its x-stencils from the neighbouring lanes (DPP wave shifts)
          instead of misaligned LDS reads: chosen when the plan's deal of groups to lanes allows it */
       MIFWI_EL_KERNEL_FWD_LANE_HALO = 16, MIFWI_EL_KERNEL_ADJ_LANE_HALO = 32 };

typedef struct mifwi_elastic_plan mifwi_elastic_plan;

int mifwi_elastic_plan_create(mifwi_elastic_plan **plan, int device,
                              const mifwi_elastic_desc *desc);
int mifwi_elastic_plan_destroy(mifwi_elastic_plan *plan);
int mifwi_elastic_plan_layout(const mifwi_elastic_plan *plan, mifwi_elastic_layout *out);
/* How the per-step kernels of this plan sweep the time range: shots per forward pass, shot groups per adjoint
 * pass (Infinity Cache residency; nshot / ngroups when everything fits or nothing does).  Introspection only. */
int mifwi_elastic_plan_pass_sizes(const mifwi_elastic_plan *plan, int32_t *forward_shots, int32_t *adjoint_groups);

/* Pressure receivers of a plan created with record_pressure = 1 (same cells and weights as the velocity
 * receivers).  The buffers stay bound to the plan until the next call of this function:
 *   rec_p [nt][nshot][nrec] or NULL: mifwi_elastic_forward writes sum w (sxx + szz)[cell], sampled after the
 *         stress update and the source term of each step (DENISE's pressure is its negative);
 *   g_p   [nt][nshot][nrec] or NULL: mifwi_elastic_backward adds w g_p[n] to the adjoint sxx and szz before the
 *         adjoint stress update of step n (the exact transpose of the sampling).                          */
int mifwi_elastic_plan_bind_pressure(mifwi_elastic_plan *plan, float *rec_p, const float *g_p);

/* Steps n = n_begin .. n_end-1.
 *   f [nt][nshot][nsrc] (added to sxx and szz - or to vx / vz, desc.source_type - pre-scaled by the host)
 *   rec_vx, rec_vz [nt][nshot][nrec] or both NULL
 *   snap NULL or [n_end-n_begin][layout.snap_step_elems]: per step and shot the five PML-filtered derivative
 *        sums the material gradient needs (exx', ezz', exz', and the two force terms); f32 format:
 *        [nshot][5][nz][gp]                                                                   */
int mifwi_elastic_forward(mifwi_elastic_plan *plan, const float *mat, const float *pz,
                          const float *px, const float *f, const int32_t *src_cell,
                          const float *src_w, const int32_t *rec_cell, const float *rec_w,
                          float *rec_vx, float *rec_vz, float *snap, float *work, int32_t n_begin,
                          int32_t n_end, int32_t flags, void *stream);

/* Adjoint steps n = n_hi down to n_lo (full run: nt-1 .. 0, ZERO_STATE|FINALIZE).
 *   g_vx, g_vz [nt][nshot][nrec] = dJ/d rec;  snapshot of step n at snap+(n-snap_first)*layout.snap_step_elems
 *   grad_mat [5][nz][gp] (on FINALIZE): dJ/d mat summed over the plan's shots
 *   grad_f NULL or [nt][nshot][nsrc]                                                        */
int mifwi_elastic_backward(mifwi_elastic_plan *plan, const float *mat, const float *pz,
                           const float *px, const int32_t *src_cell, const float *src_w,
                           const int32_t *rec_cell, const float *rec_w, const float *g_vx,
                           const float *g_vz, const float *snap, int32_t snap_first,
                           float *grad_mat, float *grad_f, float *work, int32_t n_hi, int32_t n_lo,
                           int32_t flags, void *stream);

/* Which kernel family a plan selected: the number of row slabs a shot is cut
 * into by the single-launch "cluster" time loop, or 0 when the plan runs one launch per (half)
 * step.  `adjoint` = 0 asks about the forward loop, 1 about the adjoint loop.                  */
int mifwi_acoustic_plan_cluster_slabs(const mifwi_acoustic_plan *plan, int32_t adjoint);
int mifwi_elastic_plan_cluster_slabs(const mifwi_elastic_plan *plan, int32_t adjoint);

/* ======================================================================================
 * Fused data MISFIT + adjoint source (what sits between the propagator call and .backward())
 *
 * Replaces:
 *   L1_TRACE_NORM  models/networks.py:5418-5419 (observed side), 5467-5476, 5491 (predicted side):
 *                  d = pred - direct;  dn = d / (max_t |d| + 1e-10);  loss = mean |dn - obs|
 *   L2             seisgan/fwi/layers.py:176-178; DENISE lnorm = 2 (models/networks.py:7758):
 *                  loss = 1/2 sum (pred - obs)^2
 *   GLOBAL_CORRELATION  DENISE's global-correlation norm, selectable through add_fwi_stage(lnorm=...)
 *                  (models/networks.py:9863, 10503 pass lnorm): loss = - sum_traces <pred, obs> / (|pred| |obs|)
 * pred, obs, direct (NULL = none; L1 only), adj_out (NULL = loss only): [nt][ntrace] with
 * trace = shot*nrec + receiver, the propagators' own output layout.  adj_out = dloss/dpred
 * (including the path through each trace's maximum).  loss_out: one device float.
 * work: mifwi_misfit_work_elems(kind, nt, ntrace) floats, 8-byte aligned.
 * ==================================================================================== */
enum { MIFWI_MISFIT_L1_TRACE_NORM = 0, MIFWI_MISFIT_L2 = 1, MIFWI_MISFIT_GLOBAL_CORRELATION = 2 };

int64_t mifwi_misfit_work_elems(int32_t kind, int64_t nt, int64_t ntrace);
int mifwi_misfit(int device, int32_t kind, const float *pred, const float *obs, const float *direct,
                 int64_t nt, int64_t ntrace, float *loss_out, float *adj_out, float *work, void *stream);

/* ======================================================================================
 * GRADIENT CONDITIONING on the device (what sits between d.get_fwi_gradients(...) and fake_Vp.backward(grad))
 *
 * Replaces the numpy / scipy post-processing of the model gradients in the elastic prop() methods:
 *   models/networks.py:7808-7862, 9884-9919  flipud, zero rows 0:25, scale by max(model)/max(gradient), rho x 0.1
 *   models/networks.py:10522-10540           flipud, scipy.ndimage.gaussian_filter(sigma=3), zero rows 0:5, same scaling
 *   models/networks.py:7731, 9832-9833       SWS_TAPER_GRAD_HOR / EXP_TAPER_GRAD_HOR (a weight per depth row)
 * grad, out [nplane][nz][nx] device (nplane <= 4, out != grad); models NULL or [nplane][nz][nx] device;
 * row_weight NULL or [nz] device (indexed by the row of `grad` as stored);  flip: output row j reads stored row
 * nz-1-j;  sigma: 0 = no smoothing, else scipy's Gaussian (radius int(4 sigma + 0.5) <= 32, 'reflect' boundary);
 * mute_rows: output rows [0, mute_rows) are zeroed after the smoothing;  factors NULL or HOST array [nplane].
 *   out_k = factor_k * (models ? max(models_k) / max(t_k) : 1) * t_k,   t_k = mute(smooth(w * flip(grad_k)))
 * work: mifwi_gradient_condition_work_elems(nplane) floats.
 * ==================================================================================== */
int64_t mifwi_gradient_condition_work_elems(int32_t nplane);
int mifwi_gradient_condition(int device, const float *grad, const float *models, float *out, int32_t nplane,
                             int32_t nz, int32_t nx, const float *row_weight, float sigma, int32_t flip,
                             int32_t mute_rows, const float *factors, float *work, void *stream);

/* ======================================================================================
 * MATERIAL PARAMETERISATION of the elastic kernels and its chain rule
 *
 * (Vp, Vs, rho) -> mat[5][nz][nx] = lambda dt/h, (lambda + 2 mu) dt/h, mu_xz dt/h (harmonic mean of the four mu
 * around the sxz node, 0 in water), dt/(h rho_x), dt/(h rho_z) (arithmetic means at the vx / vz nodes), edge values
 * replicated; free_surface: row 0 in the effective form of the stress-imaging condition.  What DENISE does inside
 * set_model before a forward run and undoes when get_fwi_gradients returns Vp / Vs / rho gradients
 * (models/networks.py:7698-7712, 7802-7806, 9790-9800); physicsbasedfwi2_amd/elastic.py:staggered_materials is the
 * definition (same operations and roundings).  All arrays device, [nz][nx] row-major, no padding.
 * mifwi_elastic_materials_vjp: grad_out [5][nz][nx] -> grad_vp, grad_vs, grad_rho (gather, bitwise repeatable).
 * ==================================================================================== */
int mifwi_elastic_materials(int device, const float *vp, const float *vs, const float *rho, float *mat, int32_t nz,
                            int32_t nx, float dt_over_h, int32_t free_surface, void *stream);
int mifwi_elastic_materials_vjp(int device, const float *vp, const float *vs, const float *rho, const float *grad_out,
                                float *grad_vp, float *grad_vs, float *grad_rho, int32_t nz, int32_t nx,
                                float dt_over_h, int32_t free_surface, void *stream);

/* DENISE's INVMAT1 (models/networks.py:11025 sets 2): the parameter set the gradients of `d.grad` / get_fwi_gradients
 * refer to.  Input: the gradients with respect to (Vp, Vs, rho) - what mifwi_elastic_materials_vjp returns - and the
 * model; output: the gradients with respect to
 *   MIFWI_PARAM_VELOCITY  (1)  Vp, Vs, rho                   (copy)
 *   MIFWI_PARAM_IMPEDANCE (2)  Zp = rho Vp, Zs = rho Vs, rho:  g_Zp = g_vp / rho,  g_Zs = g_vs / rho,
 *                              g_rho' = g_rho - (Vp g_vp + Vs g_vs) / rho
 *   MIFWI_PARAM_LAME      (3)  lambda, mu, rho:               g_lambda = g_vp / (2 rho Vp),
 *                              g_mu = g_vp / (rho Vp) + g_vs / (2 rho Vs)   (second term 0 where Vs = 0: a fluid has no mu to move),
 *                              g_rho' = g_rho - (Vp g_vp + Vs g_vs) / (2 rho)
 * i.e. the exact chain rule of the change of variables, cell by cell (a 3 x 3 Jacobian).  out_* may alias the inputs.
 * n = number of cells; all arrays device. */
#define MIFWI_PARAM_VELOCITY 1
#define MIFWI_PARAM_IMPEDANCE 2
#define MIFWI_PARAM_LAME 3
int mifwi_elastic_gradient_parametrization(int device, int32_t parametrization, const float *vp, const float *vs,
                                           const float *rho, const float *grad_vp, const float *grad_vs,
                                           const float *grad_rho, float *out_a, float *out_b, float *out_rho,
                                           int64_t n, void *stream);

/* vp [nz][nx] (m/s) -> r [nz + 2 pad][nx + 2 pad] = (vp dt/h)^2 with the model edge-replicated into the absorbing
 * layer: the coefficient of the scalar scheme as the deepwave-shaped call protocol needs it per call
 * (deepwave.scalar.Propagator({'vp': model}, dx), models/networks.py:5408-5411, 5449), and its chain rule
 * (grad_r [nz + 2 pad][nx + 2 pad] -> grad_vp [nz][nx]; the layer's cells folded into the edge cells in a fixed order). */
int mifwi_acoustic_coefficients(int device, const float *vp, float *r, int32_t nz, int32_t nx, int32_t pad,
                                float dt_over_h, void *stream);
int mifwi_acoustic_coefficients_vjp(int device, const float *vp, const float *grad_r, float *grad_vp, int32_t nz,
                                    int32_t nx, int32_t pad, float dt_over_h, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MIFWI_H */
