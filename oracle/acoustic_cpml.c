/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
 *
 * The scalar scheme of oracle/acoustic.c with a convolutional PML (C-PML) in place of the sponge: what
 * `deepwave.scalar.Propagator(..., pml_width=W)` is for the reference (models/networks.py:5408-5411, 5449 - a PML
 * propagator).  deepwave itself is absent from /root/reference and from
 * this image, so its own PML arithmetic is UNPINNED; this file states the published
 * second-order C-PML (Komatitsch & Martin 2007 recursive convolution, applied to the second-order wave equation as
 * in Pasalic & McGarry 2010) and is pinned by properties: reflection against the sponge, exact transposed adjoint,
 * Taylor remainder (tests/test_acoustic_cpml_oracle.py).
 *
 *   u_tt = vp^2 sum_d [ d_d( d_d u + psi_d ) + zeta_d ]        (kappa = 1)
 *   psi_d  <- b_d psi_d  + a_d  d_d u
 *   zeta_d <- b_d zeta_d + a_d (d_d^2 u + d_d psi_d)
 * a_d, b_d: the 1-D C-PML profiles at integer nodes (rows PA, PB of oracle.helpers.cpml_profiles), zero outside
 * the layer.  Discrete form on unit-spacing operators (Psi = h_d psi, Z = h_d^2 zeta; r, c_d as in acoustic.c):
 *   D1 f(i) = F1 (f(i+1) - f(i-1)) + F2 (f(i+2) - f(i-2)),  F1 = 2/3, F2 = -1/12     (4th order, as D2)
 *   phase 1:  Psi_d = fma(b_d, Psi_d, a_d * D1_d u)                                    (cells of strip d)
 *   phase 2:  Z_d   = fma(b_d, Z_d,   a_d * (D2_d u + D1_d Psi_d))                     (cells of strip d)
 *             E_d   = D1_d Psi_d + Z_d                     (non-zero on strip d and two cells beyond it)
 *   update:   u+    = fma(r, lap + fma(c0, E_0, c1 * E_1), fma(-1, u-, 2 u))  + sum_taps w f[n] r[cell]
 *             G^n   = lap + E (+ w f at source cells)  = d u^{n+1} / d r
 * Away from the strips E is an exact zero and the update is bit for bit the undamped one of acoustic.c.
 *
 * Adjoint = exact transpose, in the variable z = r * lambda of acoustic.c (one recursion serves both):
 *   w = z^{k+1};   A_d = Zb_d + c_d w;  P_d = a_d A_d;  Zb_d = b_d A_d                  (strip d)
 *   T_d = Pb_d - D1_d(c_d w + P_d);     Q_d = a_d T_d;  Pb_d = b_d T_d                  (strip d)
 *   z^k = fma(r, lap(w) + ((D2_0 P_0 - D1_0 Q_0) + (D2_1 P_1 - D1_1 Q_1)), fma(-1, z^{k+2}, 2 w)) + r R^T g^k
 *   grad_r = (1/r) sum_k z^k G^{k-1};   grad_f[k-1] = sum_taps w z^k.
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
#define F1 ((real)(2.0 / 3.0))
#define F2 ((real)(-1.0 / 12.0))
#define HALO 2

typedef struct {
    int n0, n1, nt, nshot, nsrc, nrec, ntap;
    real c0, c1;
} oracle_acoustic_cfg;          /* same layout as in acoustic.c */

typedef struct {
    const oracle_acoustic_cfg *c;
    size_t p, npad;
} geom;

static inline size_t at(const geom *g, int i0, int i1) { return (size_t)(i0 + HALO) * g->p + (size_t)(i1 + HALO); }

#define D1(f, k, s) FMA(F1, (f)[(k) + (s)] - (f)[(k) - (s)], F2 * ((f)[(k) + 2 * (s)] - (f)[(k) - 2 * (s)]))
#define D2(f, k, s) FMA(K1, (f)[(k) - (s)] + (f)[(k) + (s)], FMA(K2, (f)[(k) - 2 * (s)] + (f)[(k) + 2 * (s)], K0 * (f)[(k)]))

static inline int in_strip(const real *ab, int n, int i) { return ab[i] != 0 || ab[n + i] != 0; }

typedef struct { real *P0, *P1, *Z0, *Z1, *E; } pml_state;      /* all padded; E = scratch for the step */

static int pml_alloc(pml_state *s, size_t npad)
{
    real **f[5] = {&s->P0, &s->P1, &s->Z0, &s->Z1, &s->E};
    for (int k = 0; k < 5; ++k) { *f[k] = (real *)calloc(npad, sizeof(real)); if (!*f[k]) return 1; }
    return 0;
}
static void pml_free(pml_state *s) { free(s->P0); free(s->P1); free(s->Z0); free(s->Z1); free(s->E); }

/* forward phases 1 + 2: updates Psi, Z from u and leaves E = fma(c0, E_0, c1 E_1) in s->E */
static void pml_forward(const geom *g, const real *ab0, const real *ab1, const real *u, pml_state *s)
{
    const oracle_acoustic_cfg *c = g->c;
    const int n0 = c->n0, n1 = c->n1;
    const size_t p = g->p;
    for (int i0 = 0; i0 < n0; ++i0)
        for (int i1 = 0; i1 < n1; ++i1) {
            const size_t k = at(g, i0, i1);
            if (in_strip(ab0, n0, i0)) s->P0[k] = FMA(ab0[n0 + i0], s->P0[k], ab0[i0] * D1(u, k, p));
            if (in_strip(ab1, n1, i1)) s->P1[k] = FMA(ab1[n1 + i1], s->P1[k], ab1[i1] * D1(u, k, 1));
        }
    for (int i0 = 0; i0 < n0; ++i0)
        for (int i1 = 0; i1 < n1; ++i1) {
            const size_t k = at(g, i0, i1);
            const real dp0 = D1(s->P0, k, p), dp1 = D1(s->P1, k, 1);
            if (in_strip(ab0, n0, i0)) s->Z0[k] = FMA(ab0[n0 + i0], s->Z0[k], ab0[i0] * (D2(u, k, p) + dp0));
            if (in_strip(ab1, n1, i1)) s->Z1[k] = FMA(ab1[n1 + i1], s->Z1[k], ab1[i1] * (D2(u, k, 1) + dp1));
            const real e0 = dp0 + s->Z0[k], e1 = dp1 + s->Z1[k];
            s->E[k] = FMA(c->c0, e0, c->c1 * e1);
        }
}

/* u+ overwrites up; E = the extra term of this step (padded); G optional */
static void step(const geom *g, const real *r, const real *u, real *up, const real *E, real *G)
{
    const oracle_acoustic_cfg *c = g->c;
    const int n0 = c->n0, n1 = c->n1;
    const size_t p = g->p;
    for (int i0 = 0; i0 < n0; ++i0)
        for (int i1 = 0; i1 < n1; ++i1) {
            const size_t k = at(g, i0, i1);
            const real uc = u[k];
            const real l0 = FMA(K1, u[k - p] + u[k + p], FMA(K2, u[k - 2 * p] + u[k + 2 * p], K0 * uc));
            const real l1 = FMA(K1, u[k - 1] + u[k + 1], FMA(K2, u[k - 2] + u[k + 2], K0 * uc));
            const real lap = FMA(c->c0, l0, c->c1 * l1) + E[k];
            up[k] = FMA(r[(size_t)i0 * n1 + i1], lap, FMA((real)-1, up[k], (real)2 * uc));
            if (G) G[(size_t)i0 * n1 + i1] = lap;
        }
}

/* ab0 [2][n0] = a, b along axis 0; ab1 [2][n1].  Other arguments as oracle_acoustic_forward. */
int oracle_acoustic_cpml_forward(const oracle_acoustic_cfg *c, const real *r, const real *ab0, const real *ab1,
                                 const real *f, const int *src_cell, const real *src_w, const int *rec_cell,
                                 const real *rec_w, real *rec_out, real *G)
{
    const int n1 = c->n1, ns = c->nshot;
    const size_t ncell = (size_t)c->n0 * n1;
    geom g = {c, (size_t)(n1 + 2 * HALO), 0};
    g.npad = (size_t)(c->n0 + 2 * HALO) * g.p;
    int status = 0;
#pragma omp parallel for schedule(dynamic)
    for (int s = 0; s < ns; ++s) {
        real *ua = (real *)calloc(g.npad, sizeof(real)), *ub = (real *)calloc(g.npad, sizeof(real));
        pml_state ps;
        memset(&ps, 0, sizeof(ps));
        if (!ua || !ub || pml_alloc(&ps, g.npad)) { status = 1; free(ua); free(ub); pml_free(&ps); continue; }
        real *ucur = ua, *uprev = ub;
        for (int n = 0; n < c->nt; ++n) {
            for (int ir = 0; ir < c->nrec; ++ir) {
                real acc = 0;
                for (int t = 0; t < c->ntap; ++t) {
                    const size_t e = ((size_t)s * c->nrec + ir) * c->ntap + t;
                    const int cell = rec_cell[e];
                    if (cell >= 0) acc = FMA(rec_w[e], ucur[at(&g, cell / n1, cell % n1)], acc);
                }
                rec_out[((size_t)n * ns + s) * c->nrec + ir] = acc;
            }
            real *Gn = G ? G + ((size_t)n * ns + s) * ncell : NULL;
            pml_forward(&g, ab0, ab1, ucur, &ps);
            step(&g, r, ucur, uprev, ps.E, Gn);
            for (int is = 0; is < c->nsrc; ++is) {
                const real amp = f[((size_t)n * ns + s) * c->nsrc + is];
                for (int t = 0; t < c->ntap; ++t) {
                    const size_t e = ((size_t)s * c->nsrc + is) * c->ntap + t;
                    const int cell = src_cell[e];
                    if (cell < 0) continue;
                    const real wf = src_w[e] * amp;
                    uprev[at(&g, cell / n1, cell % n1)] += wf * r[cell];
                    if (Gn) Gn[cell] += wf;
                }
            }
            real *tmp = ucur; ucur = uprev; uprev = tmp;
        }
        free(ua); free(ub); pml_free(&ps);
    }
    return status;
}

/* adjoint phases: from w = z^{k+1} update the adjoint memory variables (s->P*, s->Z* hold Pb, Zb) and leave
 * E = (D2_0 P_0 - D1_0 Q_0) + (D2_1 P_1 - D1_1 Q_1) in s->E.  T0..T3: padded scratch (P_0, P_1, Q_0, Q_1). */
static void pml_adjoint(const geom *g, const real *ab0, const real *ab1, const real *w, pml_state *s, real **T)
{
    const oracle_acoustic_cfg *c = g->c;
    const int n0 = c->n0, n1 = c->n1;
    const size_t p = g->p;
    real *P0 = T[0], *P1 = T[1], *Q0 = T[2], *Q1 = T[3], *V0 = T[4], *V1 = T[5];
    for (int i0 = 0; i0 < n0; ++i0)
        for (int i1 = 0; i1 < n1; ++i1) {
            const size_t k = at(g, i0, i1);
            P0[k] = P1[k] = 0;
            if (in_strip(ab0, n0, i0)) {
                const real A = FMA(c->c0, w[k], s->Z0[k]);
                P0[k] = ab0[i0] * A;
                s->Z0[k] = ab0[n0 + i0] * A;
            }
            if (in_strip(ab1, n1, i1)) {
                const real A = FMA(c->c1, w[k], s->Z1[k]);
                P1[k] = ab1[i1] * A;
                s->Z1[k] = ab1[n1 + i1] * A;
            }
            V0[k] = FMA(c->c0, w[k], P0[k]);
            V1[k] = FMA(c->c1, w[k], P1[k]);
        }
    for (int i0 = 0; i0 < n0; ++i0)
        for (int i1 = 0; i1 < n1; ++i1) {
            const size_t k = at(g, i0, i1);
            Q0[k] = Q1[k] = 0;
            if (in_strip(ab0, n0, i0)) {
                const real Tt = s->P0[k] - D1(V0, k, p);
                Q0[k] = ab0[i0] * Tt;
                s->P0[k] = ab0[n0 + i0] * Tt;
            }
            if (in_strip(ab1, n1, i1)) {
                const real Tt = s->P1[k] - D1(V1, k, 1);
                Q1[k] = ab1[i1] * Tt;
                s->P1[k] = ab1[n1 + i1] * Tt;
            }
        }
    for (int i0 = 0; i0 < n0; ++i0)
        for (int i1 = 0; i1 < n1; ++i1) {
            const size_t k = at(g, i0, i1);
            s->E[k] = (D2(P0, k, p) - D1(Q0, k, p)) + (D2(P1, k, 1) - D1(Q1, k, 1));
        }
}

int oracle_acoustic_cpml_backward(const oracle_acoustic_cfg *c, const real *r, const real *ab0, const real *ab1,
                                  const int *src_cell, const real *src_w, const int *rec_cell, const real *rec_w,
                                  const real *gr, const real *G, real *grad_r, real *grad_f)
{
    const int n0 = c->n0, n1 = c->n1, ns = c->nshot;
    const size_t ncell = (size_t)n0 * n1;
    geom g = {c, (size_t)(n1 + 2 * HALO), 0};
    g.npad = (size_t)(n0 + 2 * HALO) * g.p;
    real *acc_all = (real *)calloc(ncell * ns, sizeof(real));
    if (!acc_all) return 1;
    int status = 0;
#pragma omp parallel for schedule(dynamic)
    for (int s = 0; s < ns; ++s) {
        real *za = (real *)calloc(g.npad, sizeof(real)), *zb = (real *)calloc(g.npad, sizeof(real));
        real *T[6] = {0, 0, 0, 0, 0, 0};
        pml_state ps;
        memset(&ps, 0, sizeof(ps));
        int bad = !za || !zb || pml_alloc(&ps, g.npad);
        for (int k = 0; k < 6; ++k) { T[k] = (real *)calloc(g.npad, sizeof(real)); if (!T[k]) bad = 1; }
        if (bad) { status = 1; free(za); free(zb); pml_free(&ps); for (int k = 0; k < 6; ++k) free(T[k]); continue; }
        real *zcur = za, *zprev = zb;
        real *acc = acc_all + (size_t)s * ncell;
        for (int k = c->nt - 1; k >= 1; --k) {
            pml_adjoint(&g, ab0, ab1, zcur, &ps, T);
            step(&g, r, zcur, zprev, ps.E, NULL);                     /* zprev <- z^k (no injection yet) */
            for (int ir = 0; ir < c->nrec; ++ir) {
                const real gv = gr[((size_t)k * ns + s) * c->nrec + ir];
                for (int t = 0; t < c->ntap; ++t) {
                    const size_t e = ((size_t)s * c->nrec + ir) * c->ntap + t;
                    const int cell = rec_cell[e];
                    if (cell < 0) continue;
                    zprev[at(&g, cell / n1, cell % n1)] += (rec_w[e] * gv) * r[cell];
                }
            }
            real *tmp = zcur; zcur = zprev; zprev = tmp;
            const real *Gk = G + ((size_t)(k - 1) * ns + s) * ncell;
            for (int i0 = 0; i0 < n0; ++i0)
                for (int i1 = 0; i1 < n1; ++i1) {
                    const size_t cidx = (size_t)i0 * n1 + i1;
                    acc[cidx] = FMA(zcur[at(&g, i0, i1)], Gk[cidx], acc[cidx]);
                }
            if (grad_f)
                for (int is = 0; is < c->nsrc; ++is) {
                    real a = 0;
                    for (int t = 0; t < c->ntap; ++t) {
                        const size_t e = ((size_t)s * c->nsrc + is) * c->ntap + t;
                        const int cell = src_cell[e];
                        if (cell < 0) continue;
                        a = FMA(src_w[e], zcur[at(&g, cell / n1, cell % n1)], a);
                    }
                    grad_f[((size_t)(k - 1) * ns + s) * c->nsrc + is] = a;
                }
        }
        if (grad_f)
            for (int is = 0; is < c->nsrc; ++is) grad_f[((size_t)(c->nt - 1) * ns + s) * c->nsrc + is] = 0;
        free(za); free(zb); pml_free(&ps);
        for (int k = 0; k < 6; ++k) free(T[k]);
    }
    for (size_t q = 0; q < ncell; ++q) {
        real a = 0;
        for (int s = 0; s < ns; ++s) a += acc_all[(size_t)s * ncell + q];
        grad_r[q] = a * ((real)1 / r[q]);
    }
    free(acc_all);
    return status;
}
