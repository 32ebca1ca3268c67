// 2-D constant-density acoustic propagator for gfx950 (MI355X): forward, snapshot save,
// exact discrete adjoint + imaging.  Memory-bound 4th-order star stencil, MFMA unused.
//
// Replaces (reference tree): deepwave.scalar.Propagator forward/backward as called at
// models/networks.py:5408-5411,5449,5464,5491 and the Devito Forward/Gradient operators of
// seisgan/fwi/pde/seismic/acoustic/operators.py:54-89,127-165.
//
// One kernel per time step does: stencil + damping + source/adjoint-source injection +
// receiver sampling (+ snapshot store | + imaging).  Design points for CDNA4:
//   * a thread owns 4 consecutive cells (one 16-B lane access; a wave row = 1 KiB coalesced)
//     and marches RZ rows down z with the 5-row z window held in registers, so every
//     field value is fetched from L2/HBM once per step (+4/(LZ*RZ) z-halo re-reads);
//   * u+ overwrites u- in place (u- is only read at the centre cell): 3 streams of 4 B/cell
//     + the coefficient = the 16 B/cell-step algorithmic traffic;
//   * shots that share a thread group reuse the coefficient registers and, in the adjoint,
//     one gradient accumulator, so the gradient read-modify-write is amortised over gs shots;
//   * sparse injection is staged through a per-tile LDS image only in the tiles that the
//     shot's point bounding box touches; receiver sampling runs in extra workgroups of the
//     same launch (it reads the field the step only reads), so a step is ONE launch.
// The arithmetic is the same explicit fmaf chain as oracle/acoustic.c (compile with
// -ffp-contract=off) so wavefields can be compared bitwise.
#include "mifwi_common.h"

#include <cstdlib>

namespace {

constexpr int kThreads = 256;
constexpr float K0 = -2.5f;
constexpr float K1 = (float)(4.0 / 3.0);
constexpr float K2 = (float)(-1.0 / 12.0);

struct AcParams {
    int n0, n1, ng, pitch, gp;
    long long shot_stride;       // floats per shot wavefield = (n0+4)*pitch
    int nshot, gs;
    float c0, c1;
    const float *r, *q0, *q1;
    const float *cur;            // u^n      [nshot][n0+4][pitch]
    float *prev;                 // u^{n-1} -> u^{n+1}
    float *G;                    // snapshot slice of this step [nshot][n0][gp] (SAVE: w, IMAGE: r)
    float *acc;                  // [ngroups][n0][gp]
    // injection into the new field
    int ninj, ntap_inj, inj_mode;    // mode 0: += a*r, G += a ; mode 1: += a*r*inv
    const int *inj_cell;
    const float *inj_w;
    const float *inj_amp;        // [nshot][ninj] for this step
    const int *inj_bbox;         // [nshot][4] i0min,i0max,i1min,i1max (max < min => none)
    // sampling of `cur`
    int nsmp, ntap_smp, smp_mode;    // mode 0: sum w*u ; mode 1: sum w*(1+q r)*u
    const int *smp_cell;
    const float *smp_w;
    float *smp_out;              // [nshot][nsmp] for this step (NULL: skip)
    int tiles_z;                 // grid.y rows that are stencil tiles; the rest sample
};

__device__ __forceinline__ float comp(const float4 &v, int c)
{
    return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w;
}

// ---- sampling workgroups (receivers in forward, source-gradient in backward) ----------------
__device__ void sample_points(const AcParams &p)
{
    if (p.smp_out == nullptr) return;
    const int nrb = (int)(gridDim.y - p.tiles_z) * (int)gridDim.x;
    const int rb = ((int)blockIdx.y - p.tiles_z) * (int)gridDim.x + (int)blockIdx.x;
    const int total = p.gs * p.nsmp;
    for (int e = rb * kThreads + (int)threadIdx.x; e < total; e += nrb * kThreads) {
        const int si = e / p.nsmp, ip = e - si * p.nsmp;
        const int s = (int)blockIdx.z * p.gs + si;
        if (s >= p.nshot) continue;
        const float *cur = p.cur + (long long)s * p.shot_stride;
        float a = 0.f;
        for (int t = 0; t < p.ntap_smp; ++t) {
            const long long ee = ((long long)s * p.nsmp + ip) * p.ntap_smp + t;
            const int cell = p.smp_cell[ee];
            if (cell < 0) continue;
            const int i0 = cell / p.n1, i1 = cell - i0 * p.n1;
            const float u = cur[(long long)(i0 + 2) * p.pitch + 4 + i1];
            float w = p.smp_w[ee];
            if (p.smp_mode == 1) {
                const float q = p.q0[i0] + p.q1[i1];
                w = w * (1.0f + q * p.r[(long long)i0 * p.gp + i1]);
            }
            a = fmaf(w, u, a);
        }
        p.smp_out[(long long)s * p.nsmp + ip] = a;
    }
}

template <int LX, int RZ, bool SAVE, bool IMAGE>
__global__ __launch_bounds__(kThreads) void ac_step(const AcParams p)
{
    constexpr int LZ = kThreads / LX;
    constexpr int TZ = LZ * RZ;
    constexpr int TX = LX * 4;
    if ((int)blockIdx.y >= p.tiles_z) {
        sample_points(p);
        return;
    }
    __shared__ float inj[TZ][TX];

    const int lx = (int)threadIdx.x % LX, lz = (int)threadIdx.x / LX;
    const int g = (int)blockIdx.x * LX + lx;
    const int tile_i0 = (int)blockIdx.y * TZ, tile_i1 = (int)blockIdx.x * TX;
    const int j0 = tile_i0 + lz * RZ;
    const bool active = (g < p.ng) && (j0 < p.n0);
    const int col = 4 + 4 * g;                      // first wavefield column of this group

    float4 rr[RZ], acc[RZ];
    float q0v[RZ];
    float4 q1v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        q1v = *reinterpret_cast<const float4 *>(p.q1 + 4 * g);
#pragma unroll
        for (int rz = 0; rz < RZ; ++rz) {
            const int j = j0 + rz;
            if (j < p.n0) {
                rr[rz] = *reinterpret_cast<const float4 *>(p.r + (long long)j * p.gp + 4 * g);
                q0v[rz] = p.q0[j];
                if (IMAGE)
                    acc[rz] = *reinterpret_cast<const float4 *>(
                        p.acc + ((long long)blockIdx.z * p.n0 + j) * p.gp + 4 * g);
            } else {
                rr[rz] = make_float4(0.f, 0.f, 0.f, 0.f);
                q0v[rz] = 0.f;
                if (IMAGE) acc[rz] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }

    for (int si = 0; si < p.gs; ++si) {
        const int s = (int)blockIdx.z * p.gs + si;
        if (s >= p.nshot) break;                    // block-uniform

        // ---- stage this shot's injection for this tile in LDS (block-uniform branch) -------
        bool has_inj = false;
        if (p.ninj > 0) {
            const int b0 = p.inj_bbox[4 * s + 0], b1 = p.inj_bbox[4 * s + 1];
            const int b2 = p.inj_bbox[4 * s + 2], b3 = p.inj_bbox[4 * s + 3];
            has_inj = (b0 < tile_i0 + TZ) && (b1 >= tile_i0) && (b2 < tile_i1 + TX) &&
                      (b3 >= tile_i1);
        }
        if (has_inj) {
            for (int e = (int)threadIdx.x; e < TZ * TX; e += kThreads) (&inj[0][0])[e] = 0.f;
            __syncthreads();
            const int total = p.ninj * p.ntap_inj;
            for (int e = (int)threadIdx.x; e < total; e += kThreads) {
                const long long ee = (long long)s * total + e;
                const int cell = p.inj_cell[ee];
                if (cell < 0) continue;
                const int i0 = cell / p.n1, i1 = cell - i0 * p.n1;
                const int t0 = i0 - tile_i0, t1 = i1 - tile_i1;
                if (t0 >= 0 && t0 < TZ && t1 >= 0 && t1 < TX) {
                    const float amp = p.inj_amp[(long long)s * p.ninj + e / p.ntap_inj];
                    atomicAdd(&inj[t0][t1], p.inj_w[ee] * amp);
                }
            }
            __syncthreads();
        }

        if (active) {
            const float *cur = p.cur + (long long)s * p.shot_stride;
            float *prev = p.prev + (long long)s * p.shot_stride;
            // z window rows j-2..j+1 for the first row; row j+2 is loaded in the loop
            float4 w0, w1, w2, w3, w4;
            {
                const float *c0p = cur + (long long)(j0 + 0) * p.pitch + col;   // row j0-2
                w0 = *reinterpret_cast<const float4 *>(c0p);
                w1 = *reinterpret_cast<const float4 *>(c0p + p.pitch);
                w2 = *reinterpret_cast<const float4 *>(c0p + 2 * (long long)p.pitch);
                w3 = *reinterpret_cast<const float4 *>(c0p + 3 * (long long)p.pitch);
            }
#pragma unroll
            for (int rz = 0; rz < RZ; ++rz) {
                const int j = j0 + rz;
                if (j < p.n0) {
                    const float *rowp = cur + (long long)(j + 2) * p.pitch + col;
                    w4 = *reinterpret_cast<const float4 *>(rowp + 2 * (long long)p.pitch);
                    const float2 L = *reinterpret_cast<const float2 *>(rowp - 2);
                    const float2 R = *reinterpret_cast<const float2 *>(rowp + 4);
                    float *prow = prev + (long long)(j + 2) * p.pitch + col;
                    const float4 up = *reinterpret_cast<const float4 *>(prow);
                    float4 Gv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (IMAGE)
                        Gv = *reinterpret_cast<const float4 *>(
                            p.G + ((long long)s * p.n0 + j) * p.gp + 4 * g);
                    const float xs[8] = {L.x, L.y, w2.x, w2.y, w2.z, w2.w, R.x, R.y};
                    float un[4], gk[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float uc = xs[c + 2];
                        const float s01 = comp(w1, c) + comp(w3, c);
                        const float s02 = comp(w0, c) + comp(w4, c);
                        const float s11 = xs[c + 1] + xs[c + 3];
                        const float s12 = xs[c] + xs[c + 4];
                        const float l0 = fmaf(K1, s01, fmaf(K2, s02, K0 * uc));
                        const float l1 = fmaf(K1, s11, fmaf(K2, s12, K0 * uc));
                        const float lap = fmaf(p.c0, l0, p.c1 * l1);
                        const float rv = comp(rr[rz], c);
                        const float q = q0v[rz] + comp(q1v, c);
                        const float qr = q * rv;
                        const float inv = 1.0f / (1.0f + qr);
                        const float upv = comp(up, c);
                        const float num = fmaf(rv, lap, fmaf(-(1.0f - qr), upv, 2.0f * uc));
                        float v = inv * num;
                        if (SAVE) gk[c] = inv * (fmaf(q, upv, lap) - q * v);
                        if (has_inj) {
                            const float a = inj[j - tile_i0][4 * lx + c];
                            if (p.inj_mode == 0) {
                                v += a * rv;
                                if (SAVE) gk[c] += a;
                            } else {
                                v += a * (rv * inv);
                            }
                        }
                        if (4 * g + c >= p.n1) v = 0.f;     // keep the right halo at zero
                        un[c] = v;
                    }
                    *reinterpret_cast<float4 *>(prow) = make_float4(un[0], un[1], un[2], un[3]);
                    if (SAVE)
                        *reinterpret_cast<float4 *>(p.G + ((long long)s * p.n0 + j) * p.gp +
                                                    4 * g) =
                            make_float4(gk[0], gk[1], gk[2], gk[3]);
                    if (IMAGE) {
                        acc[rz].x = fmaf(un[0], Gv.x, acc[rz].x);
                        acc[rz].y = fmaf(un[1], Gv.y, acc[rz].y);
                        acc[rz].z = fmaf(un[2], Gv.z, acc[rz].z);
                        acc[rz].w = fmaf(un[3], Gv.w, acc[rz].w);
                    }
                    w0 = w1; w1 = w2; w2 = w3; w3 = w4;
                }
            }
        }
        if (has_inj) __syncthreads();               // LDS image is reused by the next shot
    }

    if (IMAGE && active) {
#pragma unroll
        for (int rz = 0; rz < RZ; ++rz) {
            const int j = j0 + rz;
            if (j < p.n0)
                *reinterpret_cast<float4 *>(p.acc + ((long long)blockIdx.z * p.n0 + j) * p.gp +
                                            4 * g) = acc[rz];
        }
    }
}

// ---- per-shot bounding box of a point set ------------------------------------------------------
__global__ void points_bbox(const int *cell, int npts_per_shot, int n1, int *bbox)
{
    __shared__ int red[4][kThreads];
    const int s = blockIdx.x;
    int a0 = 0x7fffffff, a1 = -1, b0 = 0x7fffffff, b1 = -1;
    for (int e = threadIdx.x; e < npts_per_shot; e += kThreads) {
        const int c = cell[(long long)s * npts_per_shot + e];
        if (c < 0) continue;
        const int i0 = c / n1, i1 = c - i0 * n1;
        a0 = min(a0, i0); a1 = max(a1, i0); b0 = min(b0, i1); b1 = max(b1, i1);
    }
    red[0][threadIdx.x] = a0; red[1][threadIdx.x] = a1;
    red[2][threadIdx.x] = b0; red[3][threadIdx.x] = b1;
    __syncthreads();
    for (int st = kThreads / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            red[0][threadIdx.x] = min(red[0][threadIdx.x], red[0][threadIdx.x + st]);
            red[1][threadIdx.x] = max(red[1][threadIdx.x], red[1][threadIdx.x + st]);
            red[2][threadIdx.x] = min(red[2][threadIdx.x], red[2][threadIdx.x + st]);
            red[3][threadIdx.x] = max(red[3][threadIdx.x], red[3][threadIdx.x + st]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        bbox[4 * s + 0] = red[0][0]; bbox[4 * s + 1] = red[1][0];
        bbox[4 * s + 2] = red[2][0]; bbox[4 * s + 3] = red[3][0];
    }
}

// ---- grad_r = (sum over shot groups of acc) * (1 + q r)/r --------------------------------------
__global__ void ac_finalize(const float *acc, int ngroups, int n0, int n1, int gp, const float *r,
                            const float *q0, const float *q1, float *grad)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n0 * gp) return;
    const int i0 = (int)(idx / gp), i1 = (int)(idx - (long long)i0 * gp);
    float a = 0.f;
    for (int gidx = 0; gidx < ngroups; ++gidx) a += acc[(long long)gidx * n0 * gp + idx];
    float out = 0.f;
    if (i1 < n1) {
        const float rv = r[idx];
        const float q = q0[i0] + q1[i1];
        out = a * ((1.0f + q * rv) / rv);
    }
    grad[idx] = out;
}

}  // namespace

// ================================================================================================
struct mifwi_acoustic_plan {
    mifwi_acoustic_desc d;
    int device;
    int ng, gp, pitch, lx, rz, gs, ngroups;
    long long shot_stride, field_elems, coef_elems;
};

namespace {

template <int LX, bool SAVE, bool IMAGE>
void launch_rz(const mifwi_acoustic_plan *pl, dim3 grid, const AcParams &q, hipStream_t st)
{
    dim3 block(kThreads);
    switch (pl->rz) {
        case 8: hipLaunchKernelGGL((ac_step<LX, 8, SAVE, IMAGE>), grid, block, 0, st, q); break;
        case 2: hipLaunchKernelGGL((ac_step<LX, 2, SAVE, IMAGE>), grid, block, 0, st, q); break;
        case 1: hipLaunchKernelGGL((ac_step<LX, 1, SAVE, IMAGE>), grid, block, 0, st, q); break;
        default: hipLaunchKernelGGL((ac_step<LX, 4, SAVE, IMAGE>), grid, block, 0, st, q); break;
    }
}

template <bool SAVE, bool IMAGE>
void launch_step(const mifwi_acoustic_plan *pl, const AcParams &p, hipStream_t st)
{
    const int lz = kThreads / pl->lx;
    const int tiles_x = mifwi::ceil_div(pl->ng, pl->lx);
    const int tiles_z = mifwi::ceil_div(pl->d.n0, lz * pl->rz);
    AcParams q = p;
    q.tiles_z = tiles_z;
    int extra = 0;
    if (p.smp_out != nullptr && p.nsmp > 0)
        extra = mifwi::ceil_div(mifwi::ceil_div(pl->gs * p.nsmp, kThreads), tiles_x);
    dim3 grid(tiles_x, tiles_z + extra, pl->ngroups);
    switch (pl->lx) {
        case 64: launch_rz<64, SAVE, IMAGE>(pl, grid, q, st); break;
        case 32: launch_rz<32, SAVE, IMAGE>(pl, grid, q, st); break;
        default: launch_rz<16, SAVE, IMAGE>(pl, grid, q, st); break;
    }
}

int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

AcParams base_params(const mifwi_acoustic_plan *pl, const float *r, const float *q0,
                     const float *q1)
{
    AcParams p;
    memset(&p, 0, sizeof(p));
    p.n0 = pl->d.n0; p.n1 = pl->d.n1; p.ng = pl->ng; p.pitch = pl->pitch; p.gp = pl->gp;
    p.shot_stride = pl->shot_stride; p.nshot = pl->d.nshot; p.gs = pl->gs;
    p.c0 = pl->d.c0; p.c1 = pl->d.c1; p.r = r; p.q0 = q0; p.q1 = q1;
    return p;
}

}  // namespace

extern "C" {

const char *mifwi_last_error(void) { return mifwi::err_buf(); }
int mifwi_version(void) { return MIFWI_VERSION_MAJOR * 1000 + MIFWI_VERSION_MINOR; }
int mifwi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mifwi_acoustic_plan_create(mifwi_acoustic_plan **plan, int device,
                               const mifwi_acoustic_desc *d)
{
    if (!plan || !d) return mifwi::fail(MIFWI_EINVAL, "null plan/desc");
    if (d->n0 < 1 || d->n1 < 1 || d->nt < 1 || d->nshot < 1 || d->nsrc < 0 || d->nrec < 0)
        return mifwi::fail(MIFWI_EINVAL, "bad sizes n0=%d n1=%d nt=%d nshot=%d", d->n0, d->n1,
                           d->nt, d->nshot);
    if (d->ntap != 1 && d->ntap != 4) return mifwi::fail(MIFWI_EINVAL, "ntap must be 1 or 4");
    int rc = mifwi::check_device(device);
    if (rc) return rc;
    mifwi_acoustic_plan *pl = new mifwi_acoustic_plan;
    pl->d = *d;
    pl->device = device;
    pl->ng = mifwi::ceil_div(d->n1, 4);
    pl->gp = 4 * pl->ng;
    // one halo group left, interior, two spare groups right, rounded to 128-B lines
    pl->pitch = (int)mifwi::round_up64(4 * (pl->ng + 3), 32);
    pl->shot_stride = (long long)(d->n0 + 4) * pl->pitch;
    pl->field_elems = pl->shot_stride * d->nshot;
    pl->coef_elems = (long long)d->n0 * pl->gp;
    // lanes per row: widest tile whose padding waste stays small
    pl->lx = 16;
    for (int cand : {64, 32}) {
        const int padded = mifwi::ceil_div(pl->ng, cand) * cand;
        if (padded * 10 <= pl->ng * 12) { pl->lx = cand; break; }
    }
    pl->rz = 2;
    // tuning overrides (benchmarks only)
    { const int v = env_int("MIFWI_AC_LX", 0); if (v == 16 || v == 32 || v == 64) pl->lx = v; }
    { const int v = env_int("MIFWI_AC_RZ", 0); if (v == 1 || v == 2 || v == 4 || v == 8) pl->rz = v; }
    int gs = d->shots_per_group;
    if (gs <= 0) gs = env_int("MIFWI_AC_GS", 2);
    if (gs <= 0) gs = 2;
    if (gs > d->nshot) gs = d->nshot;
    pl->gs = gs;
    pl->ngroups = mifwi::ceil_div(d->nshot, gs);
    *plan = pl;
    return MIFWI_OK;
}

int mifwi_acoustic_plan_destroy(mifwi_acoustic_plan *plan)
{
    delete plan;
    return MIFWI_OK;
}

int mifwi_acoustic_plan_layout(const mifwi_acoustic_plan *pl, mifwi_acoustic_layout *out)
{
    if (!pl || !out) return mifwi::fail(MIFWI_EINVAL, "null plan/layout");
    out->gp = pl->gp;
    out->pitch = pl->pitch;
    out->ngroups = pl->ngroups;
    out->shots_per_group = pl->gs;
    out->field_elems = pl->field_elems;
    out->coef_elems = pl->coef_elems;
    const long long bbox = mifwi::round_up64(4LL * pl->d.nshot, 64);
    out->work_forward_elems = 2 * pl->field_elems + bbox;
    out->work_backward_elems = 2 * pl->field_elems + pl->ngroups * pl->coef_elems + bbox;
    return MIFWI_OK;
}

int mifwi_acoustic_forward(mifwi_acoustic_plan *pl, const float *r, const float *q0,
                           const float *q1, const float *f, const int32_t *src_cell,
                           const float *src_w, const int32_t *rec_cell, const float *rec_w,
                           float *rec_out, float *snap, float *work, int32_t n_begin,
                           int32_t n_end, int32_t flags, void *stream)
{
    if (!pl || !r || !q0 || !q1 || !work) return mifwi::fail(MIFWI_EINVAL, "null argument");
    const mifwi_acoustic_desc &d = pl->d;
    if (n_begin < 0 || n_end > d.nt || n_begin > n_end)
        return mifwi::fail(MIFWI_EINVAL, "bad step range [%d,%d) for nt=%d", n_begin, n_end, d.nt);
    if (d.nsrc > 0 && (!f || !src_cell || !src_w))
        return mifwi::fail(MIFWI_EINVAL, "sources declared but f/src_cell/src_w is null");
    if (rec_out && (!rec_cell || !rec_w))
        return mifwi::fail(MIFWI_EINVAL, "rec_out given but rec_cell/rec_w is null");
    int rc = mifwi::check_device(pl->device);
    if (rc) return rc;
    MIFWI_HIP_TRY(hipSetDevice(pl->device));
    hipStream_t st = (hipStream_t)stream;
    float *ua = work, *ub = work + pl->field_elems;
    int *bbox = reinterpret_cast<int *>(work + 2 * pl->field_elems);
    if (flags & MIFWI_ZERO_STATE)
        MIFWI_HIP_TRY(hipMemsetAsync(work, 0, sizeof(float) * 2 * pl->field_elems, st));
    if (d.nsrc > 0)
        hipLaunchKernelGGL(points_bbox, dim3(d.nshot), dim3(kThreads), 0, st, src_cell,
                           d.nsrc * d.ntap, d.n1, bbox);
    AcParams p = base_params(pl, r, q0, q1);
    p.ninj = d.nsrc; p.ntap_inj = d.ntap; p.inj_mode = 0;
    p.inj_cell = src_cell; p.inj_w = src_w; p.inj_bbox = bbox;
    p.nsmp = rec_out ? d.nrec : 0; p.ntap_smp = d.ntap; p.smp_mode = 0;
    p.smp_cell = rec_cell; p.smp_w = rec_w;
    const long long snap_step = (long long)d.nshot * pl->coef_elems;
    for (int n = n_begin; n < n_end; ++n) {
        p.cur = (n & 1) ? ub : ua;
        p.prev = (n & 1) ? ua : ub;
        p.inj_amp = f ? f + (long long)n * d.nshot * d.nsrc : nullptr;
        p.smp_out = (rec_out && d.nrec > 0) ? rec_out + (long long)n * d.nshot * d.nrec : nullptr;
        if (snap) {
            p.G = snap + (long long)(n - n_begin) * snap_step;
            launch_step<true, false>(pl, p, st);
        } else {
            launch_step<false, false>(pl, p, st);
        }
    }
    MIFWI_HIP_TRY(hipGetLastError());
    return MIFWI_OK;
}

int mifwi_acoustic_backward(mifwi_acoustic_plan *pl, const float *r, const float *q0,
                            const float *q1, const int32_t *src_cell, const float *src_w,
                            const int32_t *rec_cell, const float *rec_w, const float *grad_rec,
                            const float *snap, int32_t snap_first, float *grad_r, float *grad_f,
                            float *work, int32_t k_hi, int32_t k_lo, int32_t flags, void *stream)
{
    if (!pl || !r || !q0 || !q1 || !work || !snap || !grad_rec || !rec_cell || !rec_w)
        return mifwi::fail(MIFWI_EINVAL, "null argument");
    const mifwi_acoustic_desc &d = pl->d;
    if (k_lo < 1 || k_hi > d.nt - 1 || k_lo > k_hi + 1)
        return mifwi::fail(MIFWI_EINVAL, "bad adjoint range k=%d..%d for nt=%d", k_hi, k_lo, d.nt);
    if (snap_first > k_lo - 1)
        return mifwi::fail(MIFWI_EINVAL, "snapshots start at step %d but step %d is needed",
                           snap_first, k_lo - 1);
    if (grad_f && (!src_cell || !src_w))
        return mifwi::fail(MIFWI_EINVAL, "grad_f requested but src_cell/src_w is null");
    if ((flags & MIFWI_FINALIZE) && !grad_r)
        return mifwi::fail(MIFWI_EINVAL, "finalize requested but grad_r is null");
    int rc = mifwi::check_device(pl->device);
    if (rc) return rc;
    MIFWI_HIP_TRY(hipSetDevice(pl->device));
    hipStream_t st = (hipStream_t)stream;
    float *za = work, *zb = work + pl->field_elems;
    float *acc = work + 2 * pl->field_elems;
    int *bbox = reinterpret_cast<int *>(acc + (long long)pl->ngroups * pl->coef_elems);
    if (flags & MIFWI_ZERO_STATE)
        MIFWI_HIP_TRY(hipMemsetAsync(
            work, 0, sizeof(float) * (2 * pl->field_elems + pl->ngroups * pl->coef_elems), st));
    hipLaunchKernelGGL(points_bbox, dim3(d.nshot), dim3(kThreads), 0, st, rec_cell,
                       d.nrec * d.ntap, d.n1, bbox);
    AcParams p = base_params(pl, r, q0, q1);
    p.acc = acc;
    p.ninj = d.nrec; p.ntap_inj = d.ntap; p.inj_mode = 1;
    p.inj_cell = rec_cell; p.inj_w = rec_w; p.inj_bbox = bbox;
    const bool want_f = grad_f != nullptr && d.nsrc > 0;
    p.nsmp = want_f ? d.nsrc : 0; p.ntap_smp = d.ntap; p.smp_mode = 1;
    p.smp_cell = src_cell; p.smp_w = src_w;
    const long long snap_step = (long long)d.nshot * pl->coef_elems;
    // step k computes z^k from cur = z^{k+1}, prev = z^{k+2}; samples grad_f[k] from z^{k+1}.
    // Buffer parity is absolute in k so that a range can be resumed by a later call.
    for (int k = k_hi; k >= k_lo; --k) {
        const int par = (d.nt - 1 - k) & 1;
        p.cur = par ? zb : za;
        p.prev = par ? za : zb;
        p.inj_amp = grad_rec + (long long)k * d.nshot * d.nrec;
        p.G = const_cast<float *>(snap) + (long long)(k - 1 - snap_first) * snap_step;
        p.smp_out = want_f ? grad_f + (long long)k * d.nshot * d.nsrc : nullptr;
        launch_step<false, true>(pl, p, st);
    }
    if (flags & MIFWI_FINALIZE) {
        if (want_f) {
            // grad_f[k_lo-1] from z^{k_lo} (sampling workgroups only)
            AcParams s = p;
            const int par = (d.nt - 1 - (k_lo - 1)) & 1;
            s.cur = par ? zb : za;
            s.smp_out = grad_f + (long long)(k_lo - 1) * d.nshot * d.nsrc;
            s.tiles_z = 0;
            const int tiles_x = mifwi::ceil_div(pl->ng, pl->lx);
            const int extra =
                mifwi::ceil_div(mifwi::ceil_div(pl->gs * s.nsmp, kThreads), tiles_x);
            dim3 grid(tiles_x, extra, pl->ngroups), block(kThreads);
            hipLaunchKernelGGL((ac_step<16, 4, false, false>), grid, block, 0, st, s);
        }
        const long long ncoef = pl->coef_elems;
        hipLaunchKernelGGL(ac_finalize, dim3((unsigned)((ncoef + 255) / 256)), dim3(256), 0, st,
                           acc, pl->ngroups, d.n0, d.n1, pl->gp, r, q0, q1, grad_r);
    }
    MIFWI_HIP_TRY(hipGetLastError());
    return MIFWI_OK;
}

}  // extern "C"
