/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
 *
 * Plain-C CPU restatement of the 2-D constant-density acoustic propagator the
 * reference reaches through Devito (seisgan/fwi) and deepwave.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product (physicsbasedfwi2_amd/) never does.
 *
 * Scheme followed (reference file:line, paths relative to /root/reference):
 *   - PDE  m*u_tt - lap(u) + damp*u_t = q, centred dt / dt2, solved for u+ :
 *       seisgan/fwi/pde/seismic/acoustic/operators.py:24-51 (iso_stencil)
 *   - 4th-order (5 points per axis) Laplacian, space_order=4:
 *       operators.py:8-21 (laplacian), seisgan/fwi/layers.py:102
 *   - source term  u+[x] += w * src[t] * s^2/m[x]        : operators.py:81-82
 *   - receiver     rec[t] = sum_taps w * u[t][x]          : operators.py:85
 *   - adjoint time stepping with the damping sign flipped : operators.py:41-47,113
 *   - imaging condition (model gradient)                  : operators.py:152-153
 *     -- restated here as the EXACT discrete adjoint of the forward recursion in the
 *        variable r; Devito's `grad -= u.dt2*v` is the same gradient expressed in m
 *        (oracle_acoustic_gradient_devito below evaluates it literally; the two agree to
 *        round-off, tests/test_reference_pins.py).
 *
 * Parity status: the third-party arithmetic (Devito ~3.x / deepwave <=0.0.9) is
 * absent from /root/reference and from this image, so PARITY WITH THOSE PACKAGES
 * IS UNPINNED.  What is pinned: the numpy helpers of seisgan (damping profile,
 * Ricker, critical_dt, TimeAxis: tests/golden/seisgan_helpers.npz), the published
 * numbers of accuracy.ipynb reproduced with the notebook's own order-20 reference run
 * (RMS vs the analytical Green's function to 1.7 %, the space-order table to 0.2 %, the
 * time-convergence errors to 2-6 %), Devito's imaging condition evaluated literally
 * (equal to this file's gradient to round-off) and the Taylor gradient criterion
 * (gradient_example.py:143-146) - tests/test_reference_pins.py.
 *
 * Parametrisation (one kernel serves the seisgan- and the deepwave-shaped API):
 *   r[i0][i1] = s^2 / (m h^2) = vp^2 dt^2 / h^2      (h = reference spacing)
 *   q = q0[i0] + q1[i1] = damp * h^2 / (2 s)         (separable: model.py:6-29
 *                                                     adds the profile per side)
 *   a/m = 1 + q r,  inv = 1/(1+q r)
 *   u+ = inv * (2u - (1-q r) u- + r * L(u)) + sum_taps w f[n] r[cell]
 *   L(u) = c0 * D2_0(u) + c1 * D2_1(u),  D2 = (-5/2, 4/3, -1/12) unit spacing,
 *          c_k = (h/h_k)^2
 * Time loop n = 0..nt-1:  rec[n] = R u^n ; u^{n+1} = step(u^n, u^{n-1}) + inj(f[n]).
 *
 * The arithmetic is written as an explicit fmaf chain and must be compiled with
 * -ffp-contract=off so that the HIP kernels (same chain) can be compared bitwise.
 *
 * Build:  make -C oracle      (f32 -> liboracle_f32.so, f64 -> liboracle_f64.so)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORACLE_DOUBLE
typedef double real;
#define FMA(a, b, c) fma((a), (b), (c))
#else
typedef float real;
#define FMA(a, b, c) fmaf((a), (b), (c))
#endif

#define K0 ((real)-2.5)
#define K1 ((real)(4.0 / 3.0))
#define K2 ((real)(-1.0 / 12.0))
#define HALO 2

typedef struct {
    int n0, n1;        /* computational grid (already padded by the caller)   */
    int nt;            /* user time steps                                      */
    int nshot;
    int nsrc, nrec;    /* points per shot                                      */
    int ntap;          /* taps per point: 1 (cell) or 4 (bilinear)             */
    real c0, c1;       /* (h/h0)^2, (h/h1)^2                                   */
} oracle_acoustic_cfg;

/* padded scratch field: (n0+4) x (n1+4), zero halo */
static inline size_t pidx(const oracle_acoustic_cfg *c, int i0, int i1)
{
    return (size_t)(i0 + HALO) * (size_t)(c->n1 + 2 * HALO) + (size_t)(i1 + HALO);
}

/* one time step for one shot:  un overwrites up (in place leapfrog).
 * If G != NULL, stores the gradient kernel  G = d u^{n+1} / d r  (explicit part,
 * source term added by the caller).                                            */
static void step_shot(const oracle_acoustic_cfg *c, const real *r, const real *q0,
                      const real *q1, const real *u, real *up, real *G)
{
    const int n0 = c->n0, n1 = c->n1;
    const size_t p = (size_t)(n1 + 2 * HALO);
    for (int i0 = 0; i0 < n0; ++i0) {
        for (int i1 = 0; i1 < n1; ++i1) {
            const size_t k = pidx(c, i0, i1);
            const real uc = u[k];
            const real s01 = u[k - p] + u[k + p];
            const real s02 = u[k - 2 * p] + u[k + 2 * p];
            const real s11 = u[k - 1] + u[k + 1];
            const real s12 = u[k - 2] + u[k + 2];
            const real l0 = FMA(K1, s01, FMA(K2, s02, K0 * uc));
            const real l1 = FMA(K1, s11, FMA(K2, s12, K0 * uc));
            const real lap = FMA(c->c0, l0, c->c1 * l1);
            const real rr = r[(size_t)i0 * n1 + i1];
            const real q = q0[i0] + q1[i1];
            const real qr = q * rr;
            const real inv = (real)1 / ((real)1 + qr);
            const real upv = up[k];
            const real num = FMA(rr, lap, FMA(-((real)1 - qr), upv, (real)2 * uc));
            const real un = inv * num;
            if (G) G[(size_t)i0 * n1 + i1] = inv * (FMA(q, upv, lap) - q * un);
            up[k] = un;
        }
    }
}

/* rec_out [nt][nshot][nrec];  f [nt][nshot][nsrc];  cells are linear i0*n1+i1
 * (negative = inactive tap);  G [nt][nshot][n0][n1] or NULL.                   */
int oracle_acoustic_forward(const oracle_acoustic_cfg *c, const real *r, const real *q0,
                            const real *q1, const real *f, const int *src_cell,
                            const real *src_w, const int *rec_cell, const real *rec_w,
                            real *rec_out, real *G)
{
    const int n0 = c->n0, n1 = c->n1, ns = c->nshot;
    const size_t ncell = (size_t)n0 * n1;
    const size_t npad = (size_t)(n0 + 2 * HALO) * (n1 + 2 * HALO);
    int status = 0;
#pragma omp parallel for schedule(dynamic)
    for (int s = 0; s < ns; ++s) {
        real *ua = (real *)calloc(npad, sizeof(real));
        real *ub = (real *)calloc(npad, sizeof(real));
        if (!ua || !ub) { status = 1; free(ua); free(ub); continue; }
        real *ucur = ua, *uprev = ub;
        for (int n = 0; n < c->nt; ++n) {
            /* receivers read u^n */
            for (int ir = 0; ir < c->nrec; ++ir) {
                real acc = 0;
                for (int t = 0; t < c->ntap; ++t) {
                    const size_t e = ((size_t)s * c->nrec + ir) * c->ntap + t;
                    const int cell = rec_cell[e];
                    if (cell >= 0) acc = FMA(rec_w[e], ucur[pidx(c, cell / n1, cell % n1)], acc);
                }
                rec_out[((size_t)n * ns + s) * c->nrec + ir] = acc;
            }
            real *Gn = G ? G + ((size_t)n * ns + s) * ncell : NULL;
            step_shot(c, r, q0, q1, ucur, uprev, Gn);
            /* source injection into u^{n+1} (now in uprev) */
            for (int is = 0; is < c->nsrc; ++is) {
                const real amp = f[((size_t)n * ns + s) * c->nsrc + is];
                for (int t = 0; t < c->ntap; ++t) {
                    const size_t e = ((size_t)s * c->nsrc + is) * c->ntap + t;
                    const int cell = src_cell[e];
                    if (cell < 0) continue;
                    const real wf = src_w[e] * amp;
                    uprev[pidx(c, cell / n1, cell % n1)] += wf * r[cell];
                    if (Gn) Gn[cell] += wf;
                }
            }
            real *tmp = ucur; ucur = uprev; uprev = tmp;
        }
        free(ua); free(ub);
    }
    return status;
}

/* Exact discrete adjoint.  g [nt][nshot][nrec] = dJ/d rec.
 * State z^n = r*inv*lambda^n obeys the forward recursion run backwards:
 *   z^n = inv*(2 z^{n+1} - (1-q r) z^{n+2} + r L z^{n+1}) + r*inv*R^T g^n
 * grad_r = (1+q r)/r * sum_n z^{n+1} G^n ;  grad_f[n] = sum_taps w (1+q r) z^{n+1}.
 * grad_r [n0][n1] is OVERWRITTEN with the sum over shots (shot order 0..ns-1).  */
int oracle_acoustic_backward(const oracle_acoustic_cfg *c, const real *r, const real *q0,
                             const real *q1, const int *src_cell, const real *src_w,
                             const int *rec_cell, const real *rec_w, const real *g,
                             const real *G, real *grad_r, real *grad_f)
{
    const int n0 = c->n0, n1 = c->n1, ns = c->nshot;
    const size_t ncell = (size_t)n0 * n1;
    const size_t npad = (size_t)(n0 + 2 * HALO) * (n1 + 2 * HALO);
    real *acc_all = (real *)calloc(ncell * ns, sizeof(real));
    if (!acc_all) return 1;
    int status = 0;
#pragma omp parallel for schedule(dynamic)
    for (int s = 0; s < ns; ++s) {
        real *za = (real *)calloc(npad, sizeof(real));
        real *zb = (real *)calloc(npad, sizeof(real));
        if (!za || !zb) { status = 1; free(za); free(zb); continue; }
        real *zcur = za, *zprev = zb;   /* zcur = z^{k+1}, zprev = z^{k+2} */
        real *acc = acc_all + (size_t)s * ncell;
        for (int k = c->nt - 1; k >= 1; --k) {
            step_shot(c, r, q0, q1, zcur, zprev, NULL);      /* zprev <- z^k (no injection yet) */
            for (int ir = 0; ir < c->nrec; ++ir) {
                const real gv = g[((size_t)k * ns + s) * c->nrec + ir];
                for (int t = 0; t < c->ntap; ++t) {
                    const size_t e = ((size_t)s * c->nrec + ir) * c->ntap + t;
                    const int cell = rec_cell[e];
                    if (cell < 0) continue;
                    const int i0 = cell / n1, i1 = cell % n1;
                    const real rr = r[cell];
                    const real q = q0[i0] + q1[i1];
                    const real inv = (real)1 / ((real)1 + q * rr);
                    zprev[pidx(c, i0, i1)] += (rec_w[e] * gv) * (rr * inv);
                }
            }
            real *tmp = zcur; zcur = zprev; zprev = tmp;       /* zcur = z^k */
            const real *Gk = G + ((size_t)(k - 1) * ns + s) * ncell;
            for (int i0 = 0; i0 < n0; ++i0)
                for (int i1 = 0; i1 < n1; ++i1) {
                    const size_t cidx = (size_t)i0 * n1 + i1;
                    acc[cidx] = FMA(zcur[pidx(c, i0, i1)], Gk[cidx], acc[cidx]);
                }
            if (grad_f) {
                for (int is = 0; is < c->nsrc; ++is) {
                    real a = 0;
                    for (int t = 0; t < c->ntap; ++t) {
                        const size_t e = ((size_t)s * c->nsrc + is) * c->ntap + t;
                        const int cell = src_cell[e];
                        if (cell < 0) continue;
                        const int i0 = cell / n1, i1 = cell % n1;
                        const real q = q0[i0] + q1[i1];
                        a = FMA(src_w[e] * ((real)1 + q * r[cell]), zcur[pidx(c, i0, i1)], a);
                    }
                    grad_f[((size_t)(k - 1) * ns + s) * c->nsrc + is] = a;
                }
            }
        }
        if (grad_f)
            for (int is = 0; is < c->nsrc; ++is)
                grad_f[((size_t)(c->nt - 1) * ns + s) * c->nsrc + is] = 0;
        free(za); free(zb);
    }
    for (int i0 = 0; i0 < n0; ++i0)
        for (int i1 = 0; i1 < n1; ++i1) {
            const size_t cidx = (size_t)i0 * n1 + i1;
            real a = 0;
            for (int s = 0; s < ns; ++s) a += acc_all[(size_t)s * ncell + cidx];
            const real q = q0[i0] + q1[i1];
            grad_r[cidx] = a * (((real)1 + q * r[cidx]) / r[cidx]);
        }
    free(acc_all);
    return status;
}

/* Born / linearised modelling (seisgan/fwi/pde/seismic/acoustic/operators.py:168-207,
 * wavesolver.py:174-209): the first-order change of the seismograms for a model perturbation.
 * Differentiating the recursion above with respect to r gives the SAME recursion for du with the
 * distributed source  G^n * dr  (G^n = d u^{n+1}/d r, saved by the forward pass) and no point source:
 *   drec[n] = R du^n ;   du^{n+1} = step(du^n, du^{n-1}) + G^n dr.
 * dr [n0][n1];  G [nt][nshot][n0][n1];  drec_out [nt][nshot][nrec].
 * It is the exact transpose partner of oracle_acoustic_backward:  <J dr, g> = <dr, grad_r(g)>.     */
int oracle_acoustic_born(const oracle_acoustic_cfg *c, const real *r, const real *q0, const real *q1,
                         const real *dr, const real *G, const int *rec_cell, const real *rec_w,
                         real *drec_out)
{
    const int n0 = c->n0, n1 = c->n1, ns = c->nshot;
    const size_t ncell = (size_t)n0 * n1;
    const size_t npad = (size_t)(n0 + 2 * HALO) * (n1 + 2 * HALO);
    int status = 0;
#pragma omp parallel for schedule(dynamic)
    for (int s = 0; s < ns; ++s) {
        real *ua = (real *)calloc(npad, sizeof(real));
        real *ub = (real *)calloc(npad, sizeof(real));
        if (!ua || !ub) { status = 1; free(ua); free(ub); continue; }
        real *ucur = ua, *uprev = ub;
        for (int n = 0; n < c->nt; ++n) {
            for (int ir = 0; ir < c->nrec; ++ir) {
                real acc = 0;
                for (int t = 0; t < c->ntap; ++t) {
                    const size_t e = ((size_t)s * c->nrec + ir) * c->ntap + t;
                    const int cell = rec_cell[e];
                    if (cell >= 0) acc = FMA(rec_w[e], ucur[pidx(c, cell / n1, cell % n1)], acc);
                }
                drec_out[((size_t)n * ns + s) * c->nrec + ir] = acc;
            }
            step_shot(c, r, q0, q1, ucur, uprev, NULL);
            const real *Gn = G + ((size_t)n * ns + s) * ncell;
            for (int i0 = 0; i0 < n0; ++i0)
                for (int i1 = 0; i1 < n1; ++i1) {
                    const size_t cidx = (size_t)i0 * n1 + i1;
                    const size_t k = pidx(c, i0, i1);
                    uprev[k] = FMA(dr[cidx], Gn[cidx], uprev[k]);
                }
            real *tmp = ucur; ucur = uprev; uprev = tmp;
        }
        free(ua); free(ub);
    }
    return status;
}

/* ------------------------------------------------------------------------------------------------
 * Reference-literal pieces used ONLY to pin / bound the scheme above (never compared bitwise):
 *
 * (1) forward modelling with a space order 2M Laplacian (M = 1..10, Taylor weights as Devito's
 *     `u.laplace` at space_order = 2M), optional save of every wavefield u^n.  accuracy.ipynb cells 7
 *     and 14 use space_order=20 at h = 0.5 m as THE reference of the published space-order table.
 * (2) Devito's imaging condition `grad -= u.dt2 * v` (operators.py:152-153) with the adjoint
 *     recursion and receiver injection of operators.py:127-165 in Devito's own time indexing:
 *        for time = nt-2 .. 1:  v[time-1]  = stencil(v[time], v[time+1])        (operators.py:144)
 *                               v[time-1] += interp^T(res[time]) * s^2/m         (operators.py:155)
 *                               grad      -= (u[time+1]-2u[time]+u[time-1])/s^2 * v[time]   (:147)
 *     U_dev [nt][nshot][n0][n1] holds Devito's u[time] (= this file's u^{time-1}, u[0] = 0); the
 *     gradient is with respect to the square slowness m on the padded grid, summed over shots
 *     (layers.py:169-183 passes one `grad` Function to every shot).
 * ---------------------------------------------------------------------------------------------- */
static void fd2_weights(int M, double *c)       /* c[0..M]: centred 2nd-derivative weights, order 2M */
{
    double fM = 1.0;
    for (int i = 2; i <= M; ++i) fM *= i;
    c[0] = 0.0;
    for (int k = 1; k <= M; ++k) {
        double a = 1.0, b = 1.0;                /* (M-k)!, (M+k)! */
        for (int i = 2; i <= M - k; ++i) a *= i;
        for (int i = 2; i <= M + k; ++i) b *= i;
        c[k] = ((k & 1) ? 2.0 : -2.0) * fM * fM / ((double)k * k * a * b);
        c[0] -= 2.0 * c[k];
    }
}

int oracle_acoustic_forward_order(const oracle_acoustic_cfg *c, int order, const real *r, const real *q0,
                                  const real *q1, const real *f, const int *src_cell, const real *src_w,
                                  const int *rec_cell, const real *rec_w, real *rec_out, real *U)
{
    const int M = order / 2;
    if (order < 2 || order > 20 || (order & 1)) return 2;
    double w[11];
    fd2_weights(M, w);
    const int n0 = c->n0, n1 = c->n1, ns = c->nshot;
    const size_t ncell = (size_t)n0 * n1;
    const size_t p = (size_t)(n1 + 2 * M);
    const size_t npad = (size_t)(n0 + 2 * M) * p;
    int status = 0;
#define PIX(i0, i1) ((size_t)((i0) + M) * p + (size_t)((i1) + M))
#pragma omp parallel for schedule(dynamic)
    for (int s = 0; s < ns; ++s) {
        real *ua = (real *)calloc(npad, sizeof(real));
        real *ub = (real *)calloc(npad, sizeof(real));
        if (!ua || !ub) { status = 1; free(ua); free(ub); continue; }
        real *ucur = ua, *uprev = ub;
        for (int n = 0; n < c->nt; ++n) {
            for (int ir = 0; ir < c->nrec; ++ir) {
                real acc = 0;
                for (int t = 0; t < c->ntap; ++t) {
                    const size_t e = ((size_t)s * c->nrec + ir) * c->ntap + t;
                    const int cell = rec_cell[e];
                    if (cell >= 0) acc += rec_w[e] * ucur[PIX(cell / n1, cell % n1)];
                }
                rec_out[((size_t)n * ns + s) * c->nrec + ir] = acc;
            }
            if (U) {
                real *Un = U + ((size_t)n * ns + s) * ncell;
                for (int i0 = 0; i0 < n0; ++i0)
                    for (int i1 = 0; i1 < n1; ++i1) Un[(size_t)i0 * n1 + i1] = ucur[PIX(i0, i1)];
            }
            for (int i0 = 0; i0 < n0; ++i0)
                for (int i1 = 0; i1 < n1; ++i1) {
                    const size_t k = PIX(i0, i1);
                    const real uc = ucur[k];
                    real l0 = (real)w[0] * uc, l1 = (real)w[0] * uc;
                    for (int m = 1; m <= M; ++m) {
                        l0 += (real)w[m] * (ucur[k - m * p] + ucur[k + m * p]);
                        l1 += (real)w[m] * (ucur[k - m] + ucur[k + m]);
                    }
                    const real lap = c->c0 * l0 + c->c1 * l1;
                    const real rr = r[(size_t)i0 * n1 + i1];
                    const real qr = (q0[i0] + q1[i1]) * rr;
                    uprev[k] = ((real)2 * uc - ((real)1 - qr) * uprev[k] + rr * lap) / ((real)1 + qr);
                }
            for (int is = 0; is < c->nsrc; ++is) {
                const real amp = f[((size_t)n * ns + s) * c->nsrc + is];
                for (int t = 0; t < c->ntap; ++t) {
                    const size_t e = ((size_t)s * c->nsrc + is) * c->ntap + t;
                    const int cell = src_cell[e];
                    if (cell >= 0) uprev[PIX(cell / n1, cell % n1)] += src_w[e] * amp * r[cell];
                }
            }
            real *tmp = ucur; ucur = uprev; uprev = tmp;
        }
        free(ua); free(ub);
    }
#undef PIX
    return status;
}

int oracle_acoustic_gradient_devito(const oracle_acoustic_cfg *c, const real *r, const real *q0,
                                    const real *q1, const int *rec_cell, const real *rec_w,
                                    const real *res, const real *U_dev, real s, real h, real *grad_m)
{
    const int n0 = c->n0, n1 = c->n1, ns = c->nshot, nt = c->nt;
    const size_t ncell = (size_t)n0 * n1;
    const size_t npad = (size_t)(n0 + 2 * HALO) * (n1 + 2 * HALO);
    real *acc_all = (real *)calloc(ncell * ns, sizeof(real));
    if (!acc_all) return 1;
    int status = 0;
#pragma omp parallel for schedule(dynamic)
    for (int sh = 0; sh < ns; ++sh) {
        real *va = (real *)calloc(npad, sizeof(real));
        real *vb = (real *)calloc(npad, sizeof(real));
        if (!va || !vb) { status = 1; free(va); free(vb); continue; }
        real *vcur = va, *vnext = vb;          /* v[time], v[time+1] */
        real *acc = acc_all + (size_t)sh * ncell;
        for (int time = nt - 2; time >= 1; --time) {
            /* grad -= u.dt2[time] * v[time]   (reads v[time] only: order inside the iteration is free) */
            const real *up = U_dev + ((size_t)(time + 1) * ns + sh) * ncell;
            const real *u0 = U_dev + ((size_t)time * ns + sh) * ncell;
            const real *um = U_dev + ((size_t)(time - 1) * ns + sh) * ncell;
            for (int i0 = 0; i0 < n0; ++i0)
                for (int i1 = 0; i1 < n1; ++i1) {
                    const size_t ci = (size_t)i0 * n1 + i1;
                    acc[ci] -= (up[ci] - (real)2 * u0[ci] + um[ci]) / (s * s) * vcur[pidx(c, i0, i1)];
                }
            step_shot(c, r, q0, q1, vcur, vnext, NULL);            /* vnext <- v[time-1] */
            for (int ir = 0; ir < c->nrec; ++ir) {
                const real gv = res[((size_t)time * ns + sh) * c->nrec + ir];
                for (int t = 0; t < c->ntap; ++t) {
                    const size_t e = ((size_t)sh * c->nrec + ir) * c->ntap + t;
                    const int cell = rec_cell[e];
                    if (cell >= 0) vnext[pidx(c, cell / n1, cell % n1)] += rec_w[e] * gv * (r[cell] * h * h);
                }
            }
            real *tmp = vcur; vcur = vnext; vnext = tmp;
        }
        free(va); free(vb);
    }
    for (size_t ci = 0; ci < ncell; ++ci) {
        real a = 0;
        for (int sh = 0; sh < ns; ++sh) a += acc_all[(size_t)sh * ncell + ci];
        grad_m[ci] = a;
    }
    free(acc_all);
    return status;
}

int oracle_real_bytes(void) { return (int)sizeof(real); }
