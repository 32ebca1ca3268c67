/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
 *
 * Plain-C CPU restatement of the 2-D P-SV elastic propagator the reference reaches through
 * `pyapi_denise` -> DENISE-Black-Edition (`d.forward`, `d.grad`: models/networks.py:7787,
 * 9853-9877; parameters 7698-7731, 9790-9833).  DENISE itself is a third-party MPI/C code that
 * is NOT in /root/reference and not installed here, and the reference's tests hold no
 * vector for this path: parity with the DENISE BINARY is unpinned.  What IS pinned, independently of this file
 * (tests/test_elastic_analytic_pins.py, fp64, oracle/analytic_elastic.py): the scheme below reproduces the
 * closed-form Cagniard - de Hoop solutions of the P-SV equations -
 *   full space, explosive source and both line forces: L-inf error 1.5e-2 / 3.0e-3 / 9.4e-4 of the peak at
 *     h = 15 / 10 / 7.5 m (11-23 points per S wavelength), observed order 4.0 (FD_ORDER 2: order 1.8-2.0) -
 *     amplitudes, wave speeds, every half-step / half-cell staggering convention;
 *   Garvin's and Lamb's problems under the stress-imaging surface: second-order convergence with FD_ORDER 2;
 *     with FD_ORDER 4 a 3-7 % error ON the surface that does not shrink with h (zero velocities above z = 0,
 *     the image method as SOFI2D / DENISE apply it) and falls like h below it;
 *   source-receiver reciprocity to round-off (4e-3 next to the fourth-order surface).
 * The reference tree only fixes parameter names (FW/FPML/DAMPING/npower, FREE_SURF,
 * QUELLART/QUELLTYP, DH/DT/TIME) and the surrounding acquisition/normalisation code; the scheme
 * below is the published one DENISE implements, restated from the literature:
 *   - velocity-stress formulation on the standard staggered grid (Virieux 1986; Levander 1988),
 *     4th-order space (9/8, -1/24), 2nd-order leapfrog time;
 *   - arithmetic density averaging, harmonic shear-modulus averaging (done by the caller, in
 *     differentiable torch / numpy: this file receives the five staggered material arrays);
 *   - convolutional PML with memory variables (Komatitsch & Martin 2007) on every derivative;
 *   - explosive point source added to sxx and szz, receivers sample vx and vz;
 *   - the adjoint is the EXACT transpose of the discrete forward recursion (so the Taylor test
 *     of gradient_example.py:115-146 holds to rounding), not DENISE's continuous-adjoint
 *     correlation formulas.
 *
 * Staggering (row j = depth z_j, column i = x_i, x fastest):
 *   vx(j,i) @ (x_i+h/2, z_j)   vz(j,i) @ (x_i, z_j+h/2)
 *   sxx,szz(j,i) @ (x_i, z_j)  sxz(j,i) @ (x_i+h/2, z_j+h/2)
 * Unit-spacing derivative operators (fields are 0 outside the grid):
 *   Dp f(i) = C1 (f(i+1)-f(i)) + C2 (f(i+2)-f(i-1))      (forward, lands on i+1/2)
 *   Dm f(i) = C1 (f(i)-f(i-1)) + C2 (f(i+1)-f(i-2))      (backward, lands on i-1/2 of a half grid)
 * Material arrays are pre-scaled by dt/h by the caller:
 *   Ls = lambda dt/h, Ms = (lambda+2mu) dt/h, mus = mu_xz dt/h, bxs = dt/(h rho_x), bzs = dt/(h rho_z)
 * C-PML per derivative d:  psi <- b psi + a d ;  d' = d*ik + psi   (a=b=0, ik=1 outside the layer)
 *   1-D profile tables pz[6][nz], px[6][nx]: rows a, b, ik at integer nodes, then at half nodes.
 *
 * One time step n (n = 0..nt-1):
 *   V:  vx += bxs (Dp_x sxx ' + Dm_z sxz ') ;  vz += bzs (Dm_x sxz ' + Dp_z szz ')
 *   S:  sxx += Ms Dm_x vx ' + Ls Dm_z vz ' ; szz += Ls Dm_x vx ' + Ms Dm_z vz ' ;
 *       sxz += mus (Dp_z vx ' + Dp_x vz ')
 *   source: sxx, szz [cell] += w f[n] ;  receivers: rec_v*[n] = sum w v*[cell]  (after V)
 *   source_type 1 / 2 (DENISE QUELLTYPB 2 / 3, point force along x / z): vx (vz) [cell] += w f[n] between
 *   V and S instead; f arrives scaled by the host (dt/(h^2 rho) at the source node), so the adjoint of the
 *   injection is plain sampling of the adjoint velocity after S^T.
 *   pressure receivers (rec_p != NULL; DENISE SEISMO 2 / 4): rec_p[n] = sum w (sxx + szz)[cell] after S and the
 *   source term (the host applies DENISE's sign, p = -(sxx + syy)); their adjoint adds w g_p[n] to the adjoint
 *   sxx and szz before anything else of step n.
 * Saved per step for the gradient (S, 5 arrays): e1', e2', e3'+e4', d1'+d2', d3'+d4'.
 *
 * free_surface = 1 (DENISE FREE_SURF, networks.py:9811): row 0 is the free surface (szz = 0 there).
 *   V reads the stresses above it by odd mirroring: szz(-m) = -szz(m), sxz(-m) = -sxz(m-1), m = 1,2;
 *   S keeps szz(0,.) = 0 and updates sxx(0,.) with Ms - Ls^2/Ms (the caller passes row 0 of the
 *   material arrays already in that effective form: Ms_eff = Ms - Ls^2/Ms, Ls_eff = 0);
 *   velocities above the surface are zero.  The adjoint transposes exactly this: the adjoint of
 *   szz(0,.) is discarded in S^T, and V^T adds the mirrored scatter terms on rows 0..1.
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

/* Staggered first-derivative weights: Taylor order 4 (9/8, -1/24) or order 2 (1, 0) in the same four-point form
 * (cfg.fd_order; DENISE FD_ORDER).  File-scope: set at the head of each entry point (tests call them one at a time). */
static real gC1 = (real)(9.0 / 8.0), gC2 = (real)(-1.0 / 24.0);
#define C1 gC1
#define C2 gC2
#define HALO 2

typedef struct {
    int nz, nx, nt, nshot, nsrc, nrec, ntap;
    int free_surface;       /* 1: stress-imaging free surface on row 0 (see header) */
    int source_type;        /* 0: explosive (sxx, szz), 1: force on vx, 2: force on vz */
    int fd_order;           /* 4 (or 0) / 2 */
} oracle_elastic_cfg;

static void set_order(const oracle_elastic_cfg *c)
{
    if (c->fd_order == 2) { gC1 = (real)1; gC2 = (real)0; }
    else { gC1 = (real)(9.0 / 8.0); gC2 = (real)(-1.0 / 24.0); }
}

typedef struct {
    int nz, nx;
    size_t p;               /* padded pitch */
    size_t n;               /* padded elements */
} geom;

static inline size_t at(const geom *g, int j, int i) { return (size_t)(j + HALO) * g->p + (size_t)(i + HALO); }

#define DPX(f, k) FMA(C1, (f)[(k) + 1] - (f)[(k)], C2 * ((f)[(k) + 2] - (f)[(k) - 1]))
#define DMX(f, k) FMA(C1, (f)[(k)] - (f)[(k) - 1], C2 * ((f)[(k) + 1] - (f)[(k) - 2]))
#define DPZ(f, k, p) FMA(C1, (f)[(k) + (p)] - (f)[(k)], C2 * ((f)[(k) + 2 * (p)] - (f)[(k) - (p)]))
#define DMZ(f, k, p) FMA(C1, (f)[(k)] - (f)[(k) - (p)], C2 * ((f)[(k) + (p)] - (f)[(k) - 2 * (p)]))

enum { PA = 0, PB = 1, PK = 2, PAH = 3, PBH = 4, PKH = 5 };

typedef struct {
    real *vx, *vz, *sxx, *szz, *sxz;    /* padded */
    real *psi[8];                        /* compact nz*nx, index j*nx+i */
} state;

static int state_alloc(state *s, const geom *g)
{
    real **f[5] = {&s->vx, &s->vz, &s->sxx, &s->szz, &s->sxz};
    for (int k = 0; k < 5; ++k) { *f[k] = (real *)calloc(g->n, sizeof(real)); if (!*f[k]) return 1; }
    for (int k = 0; k < 8; ++k) {
        s->psi[k] = (real *)calloc((size_t)g->nz * g->nx, sizeof(real));
        if (!s->psi[k]) return 1;
    }
    return 0;
}

static void state_free(state *s)
{
    free(s->vx); free(s->vz); free(s->sxx); free(s->szz); free(s->sxz);
    for (int k = 0; k < 8; ++k) free(s->psi[k]);
}

#define PML(psi, a, b, ik, d) ((psi) = FMA((b), (psi), (a) * (d)), FMA((d), (ik), (psi)))

/* mat: [5][nz][nx] = Ls, Ms, mus, bxs, bzs.  Sn: NULL or [5][nz][nx] of this step. */
static void step_v(const geom *g, const real *mat, const real *pz, const real *px, state *s, real *Sn)
{
    const int nz = g->nz, nx = g->nx;
    const size_t nc = (size_t)nz * nx, p = g->p;
    const real *bxs = mat + 3 * nc, *bzs = mat + 4 * nc;
    for (int j = 0; j < nz; ++j)
        for (int i = 0; i < nx; ++i) {
            const size_t k = at(g, j, i), c = (size_t)j * nx + i;
            real d1 = DPX(s->sxx, k);
            real d2 = DMZ(s->sxz, k, p);
            real d3 = DMX(s->sxz, k);
            real d4 = DPZ(s->szz, k, p);
            const real d1p = PML(s->psi[0][c], px[PAH * nx + i], px[PBH * nx + i], px[PKH * nx + i], d1);
            const real d2p = PML(s->psi[1][c], pz[PA * nz + j], pz[PB * nz + j], pz[PK * nz + j], d2);
            const real d3p = PML(s->psi[2][c], px[PA * nx + i], px[PB * nx + i], px[PK * nx + i], d3);
            const real d4p = PML(s->psi[3][c], pz[PAH * nz + j], pz[PBH * nz + j], pz[PKH * nz + j], d4);
            const real s4 = d1p + d2p, s5 = d3p + d4p;
            s->vx[k] = FMA(bxs[c], s4, s->vx[k]);
            s->vz[k] = FMA(bzs[c], s5, s->vz[k]);
            if (Sn) { Sn[3 * nc + c] = s4; Sn[4 * nc + c] = s5; }
        }
}

static void step_s(const geom *g, const real *mat, const real *pz, const real *px, state *s, real *Sn)
{
    const int nz = g->nz, nx = g->nx;
    const size_t nc = (size_t)nz * nx, p = g->p;
    const real *Ls = mat, *Ms = mat + nc, *mus = mat + 2 * nc;
    for (int j = 0; j < nz; ++j)
        for (int i = 0; i < nx; ++i) {
            const size_t k = at(g, j, i), c = (size_t)j * nx + i;
            real e1 = DMX(s->vx, k);
            real e2 = DMZ(s->vz, k, p);
            real e3 = DPZ(s->vx, k, p);
            real e4 = DPX(s->vz, k);
            const real e1p = PML(s->psi[4][c], px[PA * nx + i], px[PB * nx + i], px[PK * nx + i], e1);
            const real e2p = PML(s->psi[5][c], pz[PA * nz + j], pz[PB * nz + j], pz[PK * nz + j], e2);
            const real e3p = PML(s->psi[6][c], pz[PAH * nz + j], pz[PBH * nz + j], pz[PKH * nz + j], e3);
            const real e4p = PML(s->psi[7][c], px[PAH * nx + i], px[PBH * nx + i], px[PKH * nx + i], e4);
            const real s3 = e3p + e4p;
            s->sxx[k] = FMA(Ms[c], e1p, FMA(Ls[c], e2p, s->sxx[k]));
            s->szz[k] = FMA(Ls[c], e1p, FMA(Ms[c], e2p, s->szz[k]));
            s->sxz[k] = FMA(mus[c], s3, s->sxz[k]);
            if (Sn) { Sn[c] = e1p; Sn[nc + c] = e2p; Sn[2 * nc + c] = s3; }
        }
}

/* f [nt][ns][nsrc]; rec_vx/rec_vz [nt][ns][nrec]; S NULL or [nt][ns][5][nz][nx] */
int oracle_elastic_forward(const oracle_elastic_cfg *c, const real *mat, const real *pz,
                           const real *px, const real *f, const int *src_cell, const real *src_w,
                           const int *rec_cell, const real *rec_w, real *rec_vx, real *rec_vz,
                           real *S, real *rec_p)
{
    if (c->free_surface && c->nz < 3) return 2;
    set_order(c);
    geom g = {c->nz, c->nx, (size_t)(c->nx + 2 * HALO), 0};
    g.n = (size_t)(c->nz + 2 * HALO) * g.p;
    const int ns = c->nshot, nx = c->nx;
    const size_t nc = (size_t)c->nz * c->nx;
    int status = 0;
#pragma omp parallel for schedule(dynamic)
    for (int s = 0; s < ns; ++s) {
        state st;
        memset(&st, 0, sizeof(st));
        if (state_alloc(&st, &g)) { status = 1; state_free(&st); continue; }
        for (int n = 0; n < c->nt; ++n) {
            real *Sn = S ? S + ((size_t)n * ns + s) * 5 * nc : NULL;
            if (c->free_surface)
                for (int i = 0; i < nx; ++i) {
                    st.szz[at(&g, -1, i)] = -st.szz[at(&g, 1, i)];
                    st.szz[at(&g, -2, i)] = c->nz > 2 ? -st.szz[at(&g, 2, i)] : 0;
                    st.sxz[at(&g, -1, i)] = -st.sxz[at(&g, 0, i)];
                    st.sxz[at(&g, -2, i)] = -st.sxz[at(&g, 1, i)];
                }
            step_v(&g, mat, pz, px, &st, Sn);
            if (c->source_type != 0)
                for (int is = 0; is < c->nsrc; ++is) {
                    const real amp = f[((size_t)n * ns + s) * c->nsrc + is];
                    for (int t = 0; t < c->ntap; ++t) {
                        const size_t e = ((size_t)s * c->nsrc + is) * c->ntap + t;
                        const int cell = src_cell[e];
                        if (cell < 0) continue;
                        const size_t k = at(&g, cell / nx, cell % nx);
                        if (c->source_type == 1) st.vx[k] += src_w[e] * amp;
                        else st.vz[k] += src_w[e] * amp;
                    }
                }
            step_s(&g, mat, pz, px, &st, Sn);
            for (int is = 0; is < c->nsrc && c->source_type == 0; ++is) {
                const real amp = f[((size_t)n * ns + s) * c->nsrc + is];
                for (int t = 0; t < c->ntap; ++t) {
                    const size_t e = ((size_t)s * c->nsrc + is) * c->ntap + t;
                    const int cell = src_cell[e];
                    if (cell < 0) continue;
                    const size_t k = at(&g, cell / nx, cell % nx);
                    const real a = src_w[e] * amp;
                    st.sxx[k] += a;
                    st.szz[k] += a;
                }
            }
            if (c->free_surface)       /* after the source term: szz(0,.) stays identically 0 */
                for (int i = 0; i < nx; ++i) st.szz[at(&g, 0, i)] = 0;
            for (int ir = 0; ir < c->nrec; ++ir) {
                real ax = 0, az = 0;
                for (int t = 0; t < c->ntap; ++t) {
                    const size_t e = ((size_t)s * c->nrec + ir) * c->ntap + t;
                    const int cell = rec_cell[e];
                    if (cell < 0) continue;
                    const size_t k = at(&g, cell / nx, cell % nx);
                    ax = FMA(rec_w[e], st.vx[k], ax);
                    az = FMA(rec_w[e], st.vz[k], az);
                }
                rec_vx[((size_t)n * ns + s) * c->nrec + ir] = ax;
                rec_vz[((size_t)n * ns + s) * c->nrec + ir] = az;
            }
            for (int ir = 0; ir < c->nrec && rec_p; ++ir) {
                real a = 0;
                for (int t = 0; t < c->ntap; ++t) {
                    const size_t e = ((size_t)s * c->nrec + ir) * c->ntap + t;
                    const int cell = rec_cell[e];
                    if (cell < 0) continue;
                    const size_t k = at(&g, cell / nx, cell % nx);
                    a = FMA(rec_w[e], st.sxx[k] + st.szz[k], a);
                }
                rec_p[((size_t)n * ns + s) * c->nrec + ir] = a;
            }
        }
        state_free(&st);
    }
    return status;
}

#define PMLT(psib, a, b, ik, db) /* db' in -> db out, psib updated */ \
    do { const real P__ = (psib) + (db); (db) = FMA((ik), (db), (a) * P__); (psib) = (b) * P__; } while (0)

/* Exact adjoint.  g_vx, g_vz [nt][ns][nrec];  grad_mat [5][nz][nx] (overwritten, sum over shots in
 * shot order);  grad_f NULL or [nt][ns][nsrc]. */
int oracle_elastic_backward(const oracle_elastic_cfg *c, const real *mat, const real *pz,
                            const real *px, const int *src_cell, const real *src_w,
                            const int *rec_cell, const real *rec_w, const real *g_vx,
                            const real *g_vz, const real *S, real *grad_mat, real *grad_f,
                            const real *g_p)
{
    if (c->free_surface && c->nz < 3) return 2;
    set_order(c);
    geom g = {c->nz, c->nx, (size_t)(c->nx + 2 * HALO), 0};
    g.n = (size_t)(c->nz + 2 * HALO) * g.p;
    const int ns = c->nshot, nx = c->nx, nz = c->nz;
    const size_t nc = (size_t)nz * nx, p = g.p;
    const real *Ls = mat, *Ms = mat + nc, *mus = mat + 2 * nc, *bxs = mat + 3 * nc, *bzs = mat + 4 * nc;
    real *acc_all = (real *)calloc(5 * nc * ns, sizeof(real));
    if (!acc_all) return 1;
    int status = 0;
#pragma omp parallel for schedule(dynamic)
    for (int s = 0; s < ns; ++s) {
        state st;
        memset(&st, 0, sizeof(st));
        real *T[4] = {0, 0, 0, 0};
        int bad = state_alloc(&st, &g);
        for (int k = 0; k < 4; ++k) { T[k] = (real *)calloc(g.n, sizeof(real)); if (!T[k]) bad = 1; }
        if (bad) { status = 1; state_free(&st); for (int k = 0; k < 4; ++k) free(T[k]); continue; }
        real *acc = acc_all + (size_t)s * 5 * nc;
        for (int n = c->nt - 1; n >= 0; --n) {
            const real *Sn = S + ((size_t)n * ns + s) * 5 * nc;
            /* a. receivers^T */
            for (int ir = 0; ir < c->nrec; ++ir) {
                const real gx = g_vx[((size_t)n * ns + s) * c->nrec + ir];
                const real gz = g_vz[((size_t)n * ns + s) * c->nrec + ir];
                for (int t = 0; t < c->ntap; ++t) {
                    const size_t e = ((size_t)s * c->nrec + ir) * c->ntap + t;
                    const int cell = rec_cell[e];
                    if (cell < 0) continue;
                    const size_t k = at(&g, cell / nx, cell % nx);
                    st.vx[k] += rec_w[e] * gx;
                    st.vz[k] += rec_w[e] * gz;
                }
            }
            for (int ir = 0; ir < c->nrec && g_p; ++ir) {          /* pressure receivers^T */
                const real gp = g_p[((size_t)n * ns + s) * c->nrec + ir];
                for (int t = 0; t < c->ntap; ++t) {
                    const size_t e = ((size_t)s * c->nrec + ir) * c->ntap + t;
                    const int cell = rec_cell[e];
                    if (cell < 0) continue;
                    const size_t k = at(&g, cell / nx, cell % nx);
                    st.sxx[k] += rec_w[e] * gp;
                    st.szz[k] += rec_w[e] * gp;
                }
            }
            /* free surface: szz(0,.) is identically 0 in the forward run, its adjoint is discarded */
            if (c->free_surface)
                for (int i = 0; i < nx; ++i) st.szz[at(&g, 0, i)] = 0;
            /* b. source^T */
            if (grad_f && c->source_type == 0)
                for (int is = 0; is < c->nsrc; ++is) {
                    real a = 0;
                    for (int t = 0; t < c->ntap; ++t) {
                        const size_t e = ((size_t)s * c->nsrc + is) * c->ntap + t;
                        const int cell = src_cell[e];
                        if (cell < 0) continue;
                        const size_t k = at(&g, cell / nx, cell % nx);
                        a = FMA(src_w[e], st.sxx[k] + st.szz[k], a);
                    }
                    grad_f[((size_t)n * ns + s) * c->nsrc + is] = a;
                }
            /* c. S^T : temporaries E1..E4 (padded, zero halo) */
            for (int j = 0; j < nz; ++j)
                for (int i = 0; i < nx; ++i) {
                    const size_t k = at(&g, j, i), cc = (size_t)j * nx + i;
                    const real sxxb = st.sxx[k], szzb = st.szz[k], sxzb = st.sxz[k];
                    acc[nc + cc] = FMA(Sn[cc], sxxb, FMA(Sn[nc + cc], szzb, acc[nc + cc]));      /* Ms */
                    acc[cc] = FMA(Sn[nc + cc], sxxb, FMA(Sn[cc], szzb, acc[cc]));                  /* Ls */
                    acc[2 * nc + cc] = FMA(Sn[2 * nc + cc], sxzb, acc[2 * nc + cc]);               /* mus */
                    real e1 = FMA(Ms[cc], sxxb, Ls[cc] * szzb);
                    real e2 = FMA(Ls[cc], sxxb, Ms[cc] * szzb);
                    real e3 = mus[cc] * sxzb;
                    real e4 = e3;
                    PMLT(st.psi[4][cc], px[PA * nx + i], px[PB * nx + i], px[PK * nx + i], e1);
                    PMLT(st.psi[5][cc], pz[PA * nz + j], pz[PB * nz + j], pz[PK * nz + j], e2);
                    PMLT(st.psi[6][cc], pz[PAH * nz + j], pz[PBH * nz + j], pz[PKH * nz + j], e3);
                    PMLT(st.psi[7][cc], px[PAH * nx + i], px[PBH * nx + i], px[PKH * nx + i], e4);
                    T[0][k] = e1; T[1][k] = e2; T[2][k] = e3; T[3][k] = e4;
                }
            for (int j = 0; j < nz; ++j)
                for (int i = 0; i < nx; ++i) {
                    const size_t k = at(&g, j, i);
                    st.vx[k] = st.vx[k] - (DPX(T[0], k) + DMZ(T[2], k, p));
                    st.vz[k] = st.vz[k] - (DPZ(T[1], k, p) + DMX(T[3], k));
                }
            /* force source^T: the adjoint velocity after S^T */
            if (grad_f && c->source_type != 0)
                for (int is = 0; is < c->nsrc; ++is) {
                    real a = 0;
                    for (int t = 0; t < c->ntap; ++t) {
                        const size_t e = ((size_t)s * c->nsrc + is) * c->ntap + t;
                        const int cell = src_cell[e];
                        if (cell < 0) continue;
                        const size_t k = at(&g, cell / nx, cell % nx);
                        a = FMA(src_w[e], c->source_type == 1 ? st.vx[k] : st.vz[k], a);
                    }
                    grad_f[((size_t)n * ns + s) * c->nsrc + is] = a;
                }
            /* d. V^T */
            for (int j = 0; j < nz; ++j)
                for (int i = 0; i < nx; ++i) {
                    const size_t k = at(&g, j, i), cc = (size_t)j * nx + i;
                    const real vxb = st.vx[k], vzb = st.vz[k];
                    acc[3 * nc + cc] = FMA(Sn[3 * nc + cc], vxb, acc[3 * nc + cc]);
                    acc[4 * nc + cc] = FMA(Sn[4 * nc + cc], vzb, acc[4 * nc + cc]);
                    real d1 = bxs[cc] * vxb, d2 = d1;
                    real d3 = bzs[cc] * vzb, d4 = d3;
                    PMLT(st.psi[0][cc], px[PAH * nx + i], px[PBH * nx + i], px[PKH * nx + i], d1);
                    PMLT(st.psi[1][cc], pz[PA * nz + j], pz[PB * nz + j], pz[PK * nz + j], d2);
                    PMLT(st.psi[2][cc], px[PA * nx + i], px[PB * nx + i], px[PK * nx + i], d3);
                    PMLT(st.psi[3][cc], pz[PAH * nz + j], pz[PBH * nz + j], pz[PKH * nz + j], d4);
                    T[0][k] = d1; T[1][k] = d2; T[2][k] = d3; T[3][k] = d4;
                }
            for (int j = 0; j < nz; ++j)
                for (int i = 0; i < nx; ++i) {
                    const size_t k = at(&g, j, i);
                    st.sxx[k] = st.sxx[k] - DMX(T[0], k);
                    st.sxz[k] = st.sxz[k] - (DPZ(T[1], k, p) + DPX(T[2], k));
                    st.szz[k] = st.szz[k] - DMZ(T[3], k, p);
                }
            if (c->free_surface)       /* transposed odd mirroring of the stresses read by V */
                for (int i = 0; i < nx; ++i) {
                    const size_t k0 = at(&g, 0, i), k1 = at(&g, 1, i);
                    st.sxz[k0] = st.sxz[k0] + FMA(C1, T[1][k0], C2 * T[1][k1]);
                    st.sxz[k1] = st.sxz[k1] + C2 * T[1][k0];
                    st.szz[k1] = st.szz[k1] + C2 * T[3][k0];
                }
        }
        state_free(&st);
        for (int k = 0; k < 4; ++k) free(T[k]);
    }
    for (size_t q = 0; q < 5 * nc; ++q) {
        real a = 0;
        for (int s = 0; s < ns; ++s) a += acc_all[(size_t)s * 5 * nc + q];
        grad_mat[q] = a;
    }
    free(acc_all);
    return status;
}
