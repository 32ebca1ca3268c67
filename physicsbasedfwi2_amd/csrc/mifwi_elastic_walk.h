// Column-walk fused adjoint step of the elastic propagator (per-step family, HBM-bound grids; included by
// mifwi_elastic.hip inside its anonymous namespace, after mifwi_elastic_fused.h).
//
//   el_adj_walk<BF16> : S^T and V^T of one adjoint step in ONE launch, the adjoint state read once and written once,
//                       WITHOUT the z halo of el_adj_fused.
//
// el_adj_fused recomputes E = C^T sigma_bar on a 24 x 72 region and v_bar' on 20 x 72 for every 16 x 64 tile (1.69x /
// 1.41x the arithmetic, 69 % more loads) and its three phases wait for one memory round trip each: it moves the ideal
// bytes and still loses to the two-launch pair.  Here a workgroup owns a COLUMN of 16 groups (64 cells = 256 bytes of
// every plane row, aligned to the 128-byte lines of the material / snapshot / accumulator planes) and walks down a
// chunk of rows, 14 rows per iteration, as a three-stage pipeline skewed by two rows per stage:
//
//   iteration k:   A  E  = C^T sigma_bar (transposed C-PML)        rows  z0 + 14k + 4 .. + 17      -> LDS
//                  B  v_bar' = v_bar - stencils(E), 5 gradients,    rows  z0 + 14k + 2 .. + 15      -> global, LDS
//                     D = B^T v_bar' (transposed C-PML)
//                  C  sigma_bar' = sigma_bar - stencils(D)          rows  z0 + 14k     .. + 13      -> global
//
// 224 lanes own one (row, group) each; 28 more lanes of the same 256-thread workgroup take the one halo group per side
// and row that the x stencils reach (A and B only) - one pass per phase, no second pass of a few lanes over a halo
// list.  The last four (two) rows of the E and D planes of an iteration are the first rows of the next one: they stay
// in LDS (a per-shot carry area, 6.9 KB), so nothing above the current rows is ever recomputed or re-read; what is
// left of the redundancy is the halo group per side (18 / 16 of the state loads, lines the neighbouring column
// fetches anyway) and one start-up iteration per chunk (rows z0 - 4 .. z0 + 3 of E, z0 - 2 .. z0 + 1 of D).
// The shots of an accumulator group are taken one after the other at each iteration, so the five gradient
// accumulators of the iteration's B rows stay in registers across them (40 / gs bytes per cell-step of
// read-modify-write, as in the two-launch form and in the same order: bit-identical gradients).
// Bytes per cell-step: state 20 in + 20 out, snapshot planes 20, accumulators 40 / gs, materials 20 / gs
//   =  90 at gs = 2, 75 at gs = 4  (two-launch form: 113-123 measured).
// Reads one copy of the adjoint state and writes the other (a neighbouring column may still need the old values of
// this one's edge groups); R^T g has been added to the input copy by el_inject_adjsrc.
// A lane keeps ONE absolute row modulo 14 through the three phases (slot rho: phase A takes row z0 + 14k + rho if
// rho >= 4, else the same slot 14 rows further down; phase B likewise with 2), so the adjoint stresses it loads for
// phase A are the ones it needs again for the gradients (B) and for the update (C): held in registers, loaded once.
// The four slots whose A row belongs to the next iteration park theirs in a per-lane LDS stash (3 KB per shot; each
// lane reads and writes only its own slot, no barrier involved).  [Re-loading them - L2 hits on paper - cost 24 B per
// cell-step of real fabric traffic: an XCD's L2 turns over once per iteration of its 64 workgroups.]
// Memory pipeline: every load is branch-free (clamped addresses, results discarded by lanes that do not own them) and
// addressed as uniform base + 32-bit lane offset, so the compiler counts outstanding loads exactly; the phase-A
// operands of the NEXT item (next shot of the group, or the first shot 14 rows further down) are requested behind
// phase B, the phase-C operands behind phase A.
// Arithmetic: stage_E, the v_bar' update, stage_D and the sigma_bar' update of el_adj_s / el_adj_v, term by term.

#ifndef MIFWI_WALK_WAVES
#define MIFWI_WALK_WAVES 2
#endif
constexpr int WTZ = 14;               // rows per iteration
constexpr int WOG = 16;               // owned groups per row (256 B of a plane row)
constexpr int kWOwn = WTZ * WOG;      // lanes 0 .. 223: one owned (row, group) each
constexpr int kWHalo = 2 * WTZ;       // lanes 224 .. 251: the halo group left / right of every row
static_assert(kWOwn + kWHalo <= kThreads, "own + halo lanes fit one workgroup");
// LDS planes (floats): main part [rows][64] then the two halo groups [rows][8]; z-stencil planes hold 4 carried + 14
// new rows, x-only planes 2 + 14.  The owned part has a 256-byte pitch: every ds_read_b128 lane group (16 lanes of
// two neighbouring rows) sees 16 distinct 16-byte slots.
constexpr int kWRZ = WTZ + 4, kWRX = WTZ + 2;
constexpr int kWPZ = kWRZ * 72, kWPX = kWRX * 72;
constexpr int kWE2 = 0, kWE3 = kWPZ, kWE1 = 2 * kWPZ, kWE4 = 2 * kWPZ + kWPX;
constexpr int kWD2 = 2 * kWPZ + 2 * kWPX, kWD4 = kWD2 + kWPZ, kWD1 = kWD2 + 2 * kWPZ, kWD3 = kWD1 + kWPX;
constexpr int kWWork = 4 * kWPZ + 4 * kWPX;
constexpr int kWCarryHalf = 12 * 72;  // E (or D) rows carried per shot: 4 + 4 + 2 + 2 rows of 64 + 8 floats
constexpr int kWStash = 3 * 4 * 64;   // per shot: sxx, szz, sxz of the four slots whose phase-A row is one iteration ahead
constexpr int kWCarry = 2 * kWCarryHalf + kWStash;
inline size_t walk_lds_bytes(int gs) { return sizeof(float) * (size_t)(kWWork + kWCarry * gs); }

// uniform base + 32-bit lane offset (in floats): the `global_* v_off, s[base]` addressing form
__device__ __forceinline__ const float *wk_at(const float *base, unsigned off)
{
    return reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + 4u * off);
}
__device__ __forceinline__ float *wk_at(float *base, unsigned off)
{
    return reinterpret_cast<float *>(reinterpret_cast<char *>(base) + 4u * off);
}

// where a lane's column lives inside a plane: centre, and the columns left / right of it (x stencils)
struct WLane { int cz, cx, cstr, lx, lstr, rx, rstr; };

__device__ __forceinline__ Row8 wk_row8(const float *plane, const WLane &L, int i)
{
    const float4 l = lds4(plane + L.lx + i * L.lstr), c = lds4(plane + L.cx + i * L.cstr), r = lds4(plane + L.rx + i * L.rstr);
    return Row8{{l.z, l.w, c.x, c.y, c.z, c.w, r.x, r.y}};
}

#ifdef MIFWI_ABLATIONS
#define WK_DBG(bit) (p.walk_dbg & (bit))      // 1: no snapshot loads, 2: no accumulator traffic, 4: no sigma_bar loads,
#define WK_STAMP(i) do { if (p.walk_trace) { const long long now_ = __builtin_amdgcn_s_memtime(); tr_[i] += now_ - tr_t; tr_t = now_; } } while (0)
#else                                          // 8: no v_bar loads, 16: no state stores, 32: no material loads
#define WK_DBG(bit) 0
#define WK_STAMP(i) do { } while (0)
#endif
template <bool BF16>
__global__ __launch_bounds__(kThreads, MIFWI_WALK_WAVES) void el_adj_walk(const ElParams p)
{
    const FdK K = p.K;
    int bx, by, bz;
    xcd_tile(p, bx, by, bz);
    if (by >= p.tiles_z) {
        sample_points<1>(p, bx, by, bz);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) float wbuf[];
    float *const work = wbuf;
    float *const carry = wbuf + kWWork;
    const unsigned fs = p.field_stride;
    const unsigned ncell = (unsigned)p.nz * p.gp;
    const int t = (int)threadIdx.x;
    const bool isH = t >= kWOwn;
    const int hq = t - kWOwn;
    const int side = hq & 1;
    const int r = isH ? min(hq >> 1, WTZ - 1) : t >> 4;              // slot; lanes 252 .. 255 idle along with slot 13's halo
    const int iA = r >= 4 ? r : r + WTZ;                             // row of the slot inside the iteration: phase A (4 .. 17),
    const int iB = r >= 2 ? r : r + WTZ;                             // phase B (2 .. 15); phase C takes row r
    const int gi = isH ? (side ? WOG : -1) : (t & 15);
    const bool lane_on = t < kWOwn + kWHalo;
    const int g0 = bx * WOG;
    const int g = g0 + gi;
    const bool col_ok = lane_on && g >= 0 && g < p.ng;
    const bool own_g = !isH && g < p.ng;
    WLane L;
    L.cstr = isH ? 8 : 64;
    L.cz = isH ? kWRZ * 64 + 4 * side : 4 * gi;
    L.cx = isH ? kWRX * 64 + 4 * side : 4 * gi;
    // left / right neighbour columns in the x-only planes; the far side of a halo lane is never part of a result
    L.lx = isH ? (side ? 4 * (WOG - 1) : L.cx) : (gi > 0 ? 4 * gi - 4 : kWRX * 64);
    L.lstr = isH ? (side ? 64 : 8) : (gi > 0 ? 64 : 8);
    L.rx = isH ? (side ? L.cx : 0) : (gi < WOG - 1 ? 4 * gi + 4 : kWRX * 64 + 4);
    L.rstr = isH ? (side ? 8 : 64) : (gi < WOG - 1 ? 64 : 8);
    const int ccar = isH ? 12 * 64 + 4 * side : 4 * gi;              // column inside a carry block, rows 64 / 8 apart
    const int z0 = by * p.walk_rows, z1 = min(p.nz, z0 + p.walk_rows);
    const int nk = (z1 - z0 + WTZ - 1) / WTZ;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef MIFWI_ABLATIONS
    long long tr_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_t = 0, tr_0 = 0;
    if (p.walk_trace) tr_0 = __builtin_amdgcn_s_memtime();
#endif
    const long long acc_group = (long long)(p.s0 / p.gs + bz) * 5 * p.splane;
    const int sfirst = p.s0 + bz * p.gs;
    const int ns = min(p.gs, p.nshot - sfirst);
    // carry row of this lane (rows 0-3 E2/D2, 4-7 E3/D4, 8-9 E1/D1, 10-11 E4/D3): plane, saved row, restored row
    const bool cp_on = lane_on && r < 12;
    const int cp_plane = r < 4 ? kWE2 : r < 8 ? kWE3 : r < 10 ? kWE1 : kWE4;
    const int cp_col = r < 8 ? L.cz : L.cx;
    const int cp_save = cp_plane + cp_col + (WTZ + (r < 8 ? (r & 3) : (r & 1))) * L.cstr;
    const int cp_rest = cp_plane + cp_col + (r < 8 ? (r & 3) : (r & 1)) * L.cstr;
    const int cp_car = ccar + r * L.cstr;

    const int gq = min(max(g, 0), p.ng - 1);                          // state and materials: own + halo groups
    const int go = min(max(g, g0), min(g0 + WOG, p.ng) - 1);          // own-only operands: a halo lane reads its neighbour's
    auto rowoff = [&](int row, int grp) { return (unsigned)(min(max(row, 0), p.nz - 1) + 2) * p.pitch + 4u + 4u * grp; };
    auto celloff = [&](int row, int grp) { return (unsigned)min(max(row, 0), p.nz - 1) * p.gp + 4u * grp; };

    // lanes without an owned cell store to a line of their own in the trash block: stores stay straight-line code, the
    // compiler counts them exactly, and no later wait for a load drains a store that was issued after it
    // State stores: the four floats behind the right-hand side pad of the lane's row (pitch >= gp + 12; the stencils read two
    // cells of the pad, nothing reads floats gp + 8 .. gp + 11) - same plane, so the store keeps its uniform base + 32-bit
    // lane offset.  Accumulator planes have no pad: p.trash.
    auto trashoff = [&](int row) { return (unsigned)(min(max(row, 0), p.nz - 1) + 2) * p.pitch + (unsigned)p.gp + 8u; };

    // operands of the first item: phase A (adjoint stresses, S^T materials) and phase B (snapshot planes, adjoint velocities)
    AdjIn nxt;
    float4 nS[5], nvx_, nvz_;
    BfPlanes npk;
    npk.ab = npk.de = mifwi_u4{0u, 0u, 0u, 0u};
    npk.c = mifwi_u2{0u, 0u};
#pragma unroll
    for (int q = 0; q < 5; ++q) nS[q] = zero4;
    auto request_B = [&](int s, unsigned scO, unsigned ooB) {
        const float *fin = p.fields + (long long)s * p.shot_stride;
        if (WK_DBG(1)) {
        } else if (!BF16) {
            const float *Sp = p.S + (long long)s * p.snap_shot;
#pragma unroll
            for (int q = 0; q < 5; ++q) nS[q] = mifwi::ldnt4(wk_at(Sp + q * (long long)p.splane, scO));
        } else {
            bf_request(p.S + (long long)s * p.snap_shot, p.splane, scO >> 2, npk);
        }
        nvx_ = nvz_ = zero4;
        if (!WK_DBG(8)) { nvx_ = ld4(wk_at(fin + F_VX * fs, ooB)); nvz_ = ld4(wk_at(fin + F_VZ * fs, ooB)); }
    };
    auto request_A = [&](int s, unsigned oo) {
        const float *fin = p.fields + (long long)s * p.shot_stride;
        if (!WK_DBG(4)) {
            nxt.a = ld4(wk_at(fin + F_SXX * fs, oo)); nxt.b = ld4(wk_at(fin + F_SZZ * fs, oo)); nxt.c = ld4(wk_at(fin + F_SXZ * fs, oo));
        }
    };
    nxt.a = nxt.b = nxt.c = nxt.m0 = nxt.m1 = nxt.m2 = zero4;
    request_A(sfirst, rowoff(z0 - WTZ + iA, gq));
    for (int k = -1; k < nk; ++k) {
        const int rowC = z0 + WTZ * k + r, rowB = z0 + WTZ * k + iB, rowA = z0 + WTZ * k + iA;
        const bool okA = col_ok && rowA >= max(0, z0 - 4) && rowA < min(p.nz, z1 + 4);
        const bool okB = col_ok && rowB >= max(0, z0 - 2) && rowB < min(p.nz, z1 + 2);
        const bool mineA = own_g && rowA >= z0 && rowA < z1;
        const bool mineB = own_g && rowB >= z0 && rowB < z1;
        const bool mineC = own_g && rowC >= z0 && rowC < z1;
        const unsigned ooA = rowoff(rowA, gq), ooA1 = rowoff(rowA + WTZ, gq), ccA = celloff(rowA, gq);
        const unsigned ooB = rowoff(rowB, gq), ccB = celloff(rowB, gq);
        const unsigned ooC = rowoff(rowC, go);
        // materials of the iteration's rows (shared by the shots of the group); those of the next iteration fly meanwhile
        AdjIn inA;
        inA.m0 = inA.m1 = inA.m2 = zero4;
        float4 bxs = zero4, bzs = zero4;
        if (!WK_DBG(32)) {
            inA.m0 = ld4(wk_at(p.mat + M_L * ncell, ccA)); inA.m1 = ld4(wk_at(p.mat + M_M * ncell, ccA));
            inA.m2 = ld4(wk_at(p.mat + M_MU * ncell, ccA));
            bxs = ld4(wk_at(p.mat + M_BX * ncell, ccB)); bzs = ld4(wk_at(p.mat + M_BZ * ncell, ccB));
        }
        float4 acc[5];
        const unsigned scO = snap_cell(p, min(max(rowB, 0), p.nz - 1), go);        // snapshot and accumulator planes: one layout
#pragma unroll
        for (int q = 0; q < 5; ++q) acc[q] = WK_DBG(2) ? zero4 : ld4(wk_at(p.acc + acc_group + (long long)q * p.splane, scO));
        for (int si = 0; si < ns; ++si) {
            const int s = sfirst + si;
            float *fout = p.fields_out + (long long)s * p.shot_stride;
            float *cs = carry + si * kWCarry;
            const bool last = si + 1 == ns;
#ifdef MIFWI_ABLATIONS
            if (p.walk_trace) { tr_t = __builtin_amdgcn_s_memtime(); tr_[6] += 1; }
#endif
            // ---- operands of phase B: requested now, used behind the first barrier; those of phase A were requested
            //      during the item before, in front of its stores ------------------------------------------------------
            request_B(s, scO, ooB);
            float4 S1 = nS[0], S2 = nS[1], S3 = nS[2], S4 = nS[3], S5 = nS[4];
            BfPlanes packed = npk;
            inA.a = nxt.a; inA.b = nxt.b; inA.c = nxt.c;
            const float4 vxb = nvx_, vzb = nvz_;
            // the adjoint stresses of this lane's B and C rows: what it has just received for phase A, or (slots 0-3)
            // what it parked an iteration ago
            float *stash = cs + 2 * kWCarryHalf + (isH ? 0 : 4 * (t & 15) + 64 * (r & 3));
            const float4 raw_a = inA.a, raw_b = inA.b, raw_c = inA.c;
            const float4 old_a = lds4(stash), old_b = lds4(stash + 4 * 64), old_c = lds4(stash + 8 * 64);
            const float4 bxx = r >= 4 || r < 2 ? raw_a : old_a, bxz = r >= 4 || r < 2 ? raw_c : old_c;
            float4 bzz = r >= 4 || r < 2 ? raw_b : old_b;
            const float4 sxx0 = r >= 4 ? raw_a : old_a, szz0 = r >= 4 ? raw_b : old_b, sxz0 = r >= 4 ? raw_c : old_c;
            if (!isH && r < 4) { sts4(stash, raw_a); sts4(stash + 4 * 64, raw_b); sts4(stash + 8 * 64, raw_c); }
            // ---- A: E of rows z0 + 14k + 4 .. + 17; the rows above them come back from the carry area -------------
            if (k >= 0 && cp_on) sts4(work + cp_rest, lds4(cs + cp_car));
            {
                if (p.fsurf && rowA == 0) inA.b = zero4;         // S^T: the adjoint of szz(0,.) is discarded
                float4 e1 = zero4, e2 = zero4, e3 = zero4, e4 = zero4;
                if (okA) stage_E(p, s, f_opaque(rowA), f_opaque(g), inA, mineA, e1, e2, e3, e4);
                if (lane_on) {
                    sts4(work + kWE2 + L.cz + iA * L.cstr, e2); sts4(work + kWE3 + L.cz + iA * L.cstr, e3);
                    sts4(work + kWE1 + L.cx + (iA - 2) * L.cstr, e1); sts4(work + kWE4 + L.cx + (iA - 2) * L.cstr, e4);
                }
            }
            WK_STAMP(0);
            __syncthreads();
            WK_STAMP(1);
            // ---- B: v_bar', gradients, D of rows z0 + 14k + 2 .. + 15 ------------------------------------------------
            if (k >= 0 && cp_on) sts4(work + kWD2 + cp_rest, lds4(cs + kWCarryHalf + cp_car));
            float4 nx4, nz4;
            {
                if (p.fsurf && rowB == 0) bzz = zero4;
                if (BF16) {
                    bf_pin(packed);
                    bf_widen(packed, S1, S2, S3, S4, S5);
                }
                // gradients (oracle order): Ms, Ls, mus from sigma_bar; bxs, bzs from the new v_bar.  Lanes that own no
                // cell of the B rows go through the same arithmetic on whatever they hold; their sums are never stored.
#define ACC3(dst, a, b, c_, d) dst = fmaf(a, b, fmaf(c_, d, dst))
                ACC3(acc[M_M].x, S1.x, bxx.x, S2.x, bzz.x); ACC3(acc[M_M].y, S1.y, bxx.y, S2.y, bzz.y);
                ACC3(acc[M_M].z, S1.z, bxx.z, S2.z, bzz.z); ACC3(acc[M_M].w, S1.w, bxx.w, S2.w, bzz.w);
                ACC3(acc[M_L].x, S2.x, bxx.x, S1.x, bzz.x); ACC3(acc[M_L].y, S2.y, bxx.y, S1.y, bzz.y);
                ACC3(acc[M_L].z, S2.z, bxx.z, S1.z, bzz.z); ACC3(acc[M_L].w, S2.w, bxx.w, S1.w, bzz.w);
#undef ACC3
                acc[M_MU].x = fmaf(S3.x, bxz.x, acc[M_MU].x); acc[M_MU].y = fmaf(S3.y, bxz.y, acc[M_MU].y);
                acc[M_MU].z = fmaf(S3.z, bxz.z, acc[M_MU].z); acc[M_MU].w = fmaf(S3.w, bxz.w, acc[M_MU].w);
                __builtin_amdgcn_sched_barrier(0);        // S1-S3 and the B rows' stresses are dead before the E planes are read
                float nvx[4], nvz[4];
                {
                    const int lr = iB - 2;
                    const Row8 x1 = wk_row8(work + kWE1, L, lr), x4 = wk_row8(work + kWE4, L, lr);
                    const float *E3 = work + kWE3 + L.cz, *E2 = work + kWE2 + L.cz;
                    const float4 z3a = lds4(E3 + (lr + 0) * L.cstr), z3b = lds4(E3 + (lr + 1) * L.cstr);
                    const float4 z3c = lds4(E3 + (lr + 2) * L.cstr), z3d = lds4(E3 + (lr + 3) * L.cstr);
                    const float4 z2a = lds4(E2 + (lr + 1) * L.cstr), z2b = lds4(E2 + (lr + 2) * L.cstr);
                    const float4 z2c = lds4(E2 + (lr + 3) * L.cstr), z2d = lds4(E2 + (lr + 4) * L.cstr);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float dx1 = dfw(K, x1.v[c + 1], x1.v[c + 2], x1.v[c + 3], x1.v[c + 4]);
                        const float dz3 = dbw(K, comp(z3a, c), comp(z3b, c), comp(z3c, c), comp(z3d, c));
                        const float dz2 = dfw(K, comp(z2a, c), comp(z2b, c), comp(z2c, c), comp(z2d, c));
                        const float dx4 = dbw(K, x4.v[c], x4.v[c + 1], x4.v[c + 2], x4.v[c + 3]);
                        float ax = comp(vxb, c) - (dx1 + dz3);
                        float az = comp(vzb, c) - (dz2 + dx4);
                        if (4 * g + c >= p.nx || !okB) { ax = 0.f; az = 0.f; }
                        nvx[c] = ax; nvz[c] = az;
                    }
                }
                nx4 = make_float4(nvx[0], nvx[1], nvx[2], nvx[3]); nz4 = make_float4(nvz[0], nvz[1], nvz[2], nvz[3]);
                acc[M_BX].x = fmaf(S4.x, nvx[0], acc[M_BX].x); acc[M_BX].y = fmaf(S4.y, nvx[1], acc[M_BX].y);
                acc[M_BX].z = fmaf(S4.z, nvx[2], acc[M_BX].z); acc[M_BX].w = fmaf(S4.w, nvx[3], acc[M_BX].w);
                acc[M_BZ].x = fmaf(S5.x, nvz[0], acc[M_BZ].x); acc[M_BZ].y = fmaf(S5.y, nvz[1], acc[M_BZ].y);
                acc[M_BZ].z = fmaf(S5.z, nvz[2], acc[M_BZ].z); acc[M_BZ].w = fmaf(S5.w, nvz[3], acc[M_BZ].w);
                float4 d1 = zero4, d2 = zero4, d3 = zero4, d4 = zero4;
                if (okB) stage_D(p, s, f_opaque(rowB), f_opaque(g), nx4, nz4, bxs, bzs, mineB, d1, d2, d3, d4);
                if (lane_on) {
                    sts4(work + kWD2 + L.cz + (iB + 2) * L.cstr, d2); sts4(work + kWD4 + L.cz + (iB + 2) * L.cstr, d4);
                    sts4(work + kWD1 + L.cx + iB * L.cstr, d1); sts4(work + kWD3 + L.cx + iB * L.cstr, d3);
                }
            }
            // ---- phase-A operands of the next item (the next shot on these rows, or the first shot 14 rows down) are
            //      requested in front of this phase's stores --------------------------------------------------------------
            request_A(last ? sfirst : s + 1, last ? ooA1 : ooA);
            if (!WK_DBG(16)) {
                const unsigned o = mineB ? ooB : trashoff(rowB);
                st4(wk_at(fout + F_VX * fs, o), nx4);
                st4(wk_at(fout + F_VZ * fs, o), nz4);
            }
            WK_STAMP(2);
            __syncthreads();
            WK_STAMP(3);
            // ---- C: sigma_bar' of rows z0 + 14k .. + 13; the planes' last rows go to the carry area -------------------
            if (cp_on) {
                sts4(cs + cp_car, lds4(work + cp_save));
                sts4(cs + kWCarryHalf + cp_car, lds4(work + kWD2 + cp_save));
            }
            float nxx[4], nzz[4], nxz[4];
            {
                const float *D2 = work + kWD2 + L.cz, *D4 = work + kWD4 + L.cz;
                const Row8 x1 = wk_row8(work + kWD1, L, r), x3 = wk_row8(work + kWD3, L, r);
                const float4 z2a = lds4(D2 + (r + 1) * L.cstr), z2b = lds4(D2 + (r + 2) * L.cstr);
                const float4 z2c = lds4(D2 + (r + 3) * L.cstr), z2d = lds4(D2 + (r + 4) * L.cstr);
                const float4 z4a = lds4(D4 + (r + 0) * L.cstr), z4b = lds4(D4 + (r + 1) * L.cstr);
                const float4 z4c = lds4(D4 + (r + 2) * L.cstr), z4d = lds4(D4 + (r + 3) * L.cstr);
                float4 m12 = zero4, m13 = zero4, m32 = zero4;
                if (p.fsurf && rowC < 2) {         // rows 0, 1 of the grid: k = 0 of the top chunk, plane row 2 is grid row 0
                    m12 = lds4(D2 + 2 * L.cstr); m13 = lds4(D2 + 3 * L.cstr); m32 = lds4(D4 + 2 * L.cstr);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float dx1 = dbw(K, x1.v[c], x1.v[c + 1], x1.v[c + 2], x1.v[c + 3]);
                    const float dz2 = dfw(K, comp(z2a, c), comp(z2b, c), comp(z2c, c), comp(z2d, c));
                    const float dx3 = dfw(K, x3.v[c + 1], x3.v[c + 2], x3.v[c + 3], x3.v[c + 4]);
                    const float dz4 = dbw(K, comp(z4a, c), comp(z4b, c), comp(z4c, c), comp(z4d, c));
                    nxx[c] = comp(sxx0, c) - dx1;
                    nxz[c] = comp(sxz0, c) - (dz2 + dx3);
                    nzz[c] = comp(szz0, c) - dz4;
                    if (p.fsurf && rowC < 2) {
                        if (rowC == 0) nxz[c] = nxz[c] + fmaf(K.c1, comp(m12, c), K.c2 * comp(m13, c));
                        else { nxz[c] = nxz[c] + K.c2 * comp(m12, c); nzz[c] = nzz[c] + K.c2 * comp(m32, c); }
                    }
                    if (4 * g + c >= p.nx) { nxx[c] = 0.f; nxz[c] = 0.f; nzz[c] = 0.f; }
                }
            }
            if (!WK_DBG(16)) {
                const unsigned o = mineC ? ooC : trashoff(rowC);
                st4(wk_at(fout + F_SXX * fs, o), make_float4(nxx[0], nxx[1], nxx[2], nxx[3]));
                st4(wk_at(fout + F_SZZ * fs, o), make_float4(nzz[0], nzz[1], nzz[2], nzz[3]));
                st4(wk_at(fout + F_SXZ * fs, o), make_float4(nxz[0], nxz[1], nxz[2], nxz[3]));
            }
            WK_STAMP(4);
            __syncthreads();          // the work planes are reused by the next item
            WK_STAMP(5);
        }
        if (!WK_DBG(2)) {
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                float *real = wk_at(p.acc + acc_group + (long long)q * p.splane, scO);
                st4(mineB ? real : p.trash + 4 * f_opaque(t), acc[q]);
            }
        }
    }
#ifdef MIFWI_ABLATIONS
    if (p.walk_trace && t == 0) {
        long long *o = p.walk_trace + 8 * ((long long)bx + (long long)gridDim.x * (by + (long long)p.tiles_z * bz));
        tr_[7] = __builtin_amdgcn_s_memtime() - tr_0;
        for (int i = 0; i < 8; ++i) o[i] = tr_[i];
    }
#endif
}
