// CLUSTER kernels of the elastic propagator: the whole time loop in ONE launch, the five wavefields
// of a row slab resident in LDS, memory variables / materials of a thread's cells in registers.
// Included by mifwi_elastic.hip inside its anonymous namespace (uses its helpers and enums).
//
// Why (profiles/): on the grids the reference actually runs (100x300, 150x294,
// 190x324 ...) a time step launched as kernels is bound by launch boundaries and memory round trips
// (V 12 us + S 14 us for 0.96 M cells), not by HBM.  Here a shot is cut into NW row slabs, one
// 512-thread workgroup (one CU, ~105 KB of LDS) per slab, NW x shots <= CU count so that every
// workgroup is resident; per step the only global traffic is the snapshot stream (20 B/cell) and
// the two halo hand-offs (velocities after V, stresses after S), done as self-validating 8-byte
// {epoch,value} granules with agent-scope stores/loads (cdna_hip_programming.md, G16 form R2):
// no flags, no fences, no grid barrier, every spin bounded.  The hand-off latency is hidden behind
// the rows that do not need the halo: interior rows are updated before the poll, boundary rows after.
// Arithmetic per cell = el_step_v / el_step_s (bitwise identical seismograms).
#pragma once

constexpr int kEcThreads = 512;       // 8 waves/CU: the 256-VGPR budget holds two 4-cell groups' materials and
                                      // memory variables (adjoint: fields, accumulators, snapshot terms)
constexpr unsigned kEcMaxSpin = 400000;

struct EcParams {
    int nz, nx, ng, gp, pitch;
    unsigned field_stride;
    long long shot_stride;
    int nshot, NW, PL, shot0, shot1;
    int n_first, n_last;                 // steps n_first .. n_last-1
    int W, wl, xr0, wx, fsurf;
    long long psix_shot, psiz_shot;
    const float *mat, *pz, *px;
    float *fields, *psix, *psiz;         // global state (layout of the per-step kernels)
    float *S;                            // snapshots of step n at S + (n - s_first) * s_step (or null)
    int s_first;
    long long s_step;
    int nsrc, nrec;
    const int *src_cell;
    const float *src_w, *f;              // f [nt][nshot][nsrc]
    const int *rec_cell;
    const float *rec_w;
    float *rec_vx, *rec_vz;              // [nt][nshot][nrec] or null
    unsigned long long *xbuf;            // [nshot][NW][2 kinds][2 parities][8*gp] granules
    int *err;
    int *xcc_tab;                        // [nshot][NW] XCC_ID + 1 of each slab's workgroup (mifwi::same_xcd)
    int dbg, nap;
    FdK K;                               // stencil weights (fd_order)
#ifdef MIFWI_ABLATIONS
    long long *trace;                    // phase time stamps of one workgroup (MIFWI_EL_CL_TRACE), see EC_STAMP
#endif
};

// Ablation builds: wave 0..7 of slab NW/2 of the first shot writes s_memtime at phase boundary k of steps 64..127
// to trace[((it - 64) * 8 + wave) * 16 + k]; tools/cluster_trace.py turns them into a per-phase table.
#ifdef MIFWI_ABLATIONS
#define EC_STAMP(k)                                                                                        \
    do {                                                                                                   \
        if (p.trace && tr_on && it >= 64 && it < 128 && (t & 63) == 0)                                     \
            p.trace[((it - 64) * 8 + (t >> 6)) * 16 + (k)] = (long long)__builtin_readcyclecounter();      \
    } while (0)
// dbg bit 4096 (fault injection): slab 1 of the first shot stalls ~10 ms at every 64th step - a neighbour that is late
// but not absent, what a GPU shared with another process looks like; the others must wait (no time-out), finish with
// the same results, and the launch must be counted by mifwi_slow_handoff_count()
#define EC_LAGGARD()                                                                                       \
    do {                                                                                                   \
        if ((kDbg(p) & 4096) && w == 1 && s == p.shot0 && (it & 63) == 1)                                  \
            for (int z_ = 0; z_ < 3000; ++z_) __builtin_amdgcn_s_sleep(127);                               \
    } while (0)
#else
#define EC_STAMP(k) do { } while (0)
#define EC_LAGGARD() do { } while (0)
#endif

__host__ __device__ __forceinline__ void ec_slab_rows(int nz, int NW, int w, int &r0, int &rows)
{
    const int base = nz / NW, rem = nz - base * NW;
    rows = base + (w < rem ? 1 : 0);
    r0 = w * base + (w < rem ? w : rem);
}

// One exchange slot = kEcRowFields rows of gp granules.  A z-stencil reads the backward-differenced
// field (vz, sxz; adjoint E3, D4: "A") on rows j-2..j+1 and the forward-differenced one (vx, szz;
// adjoint E2, D2: "B") on rows j-1..j+2, so a slab needs A on 2 rows above / 1 below and B on 1 row
// above / 2 below: six row-fields per hand-off, not eight.
//   row-field  field  owner's row   lands in the neighbour's LDS row
//      0         A       0           R'+2   (first bottom halo row of the slab above)
//      1         B       0           R'+2
//      2         B       1           R'+3
//      3         A       R-2         0      (top halo rows of the slab below)
//      4         A       R-1         1
//      5         B       R-1         1
// Inside a row the order is [k 0..3][group] (cell 4*group + k): the lanes of a publishing wave hold
// consecutive groups, so each of their store instructions writes whole lines (8-byte stores 32 bytes
// apart are several times slower to land).
constexpr int kEcRowFields = 6;
constexpr int kEcGr = 5;              // granules a thread receives per hand-off: 6*gp <= kEcGr*kEcThreads

__device__ __forceinline__ int ec_rf_field(int rf) { return (rf == 0 || rf == 3 || rf == 4) ? 1 : 0; }   // 1 = A
__device__ __forceinline__ int ec_rf_lds_row(int rf, int R)
{
    return rf == 0 || rf == 1 ? R + 2 : rf == 2 ? R + 3 : rf == 3 ? 0 : 1;
}

// Value the compiler must treat as unknown at this point.  Every per-lane quantity that does not change
// over the time loop (LDS offsets, group class, strip slots ...) goes through this once per use: left
// alone, the compiler hoists every address `base + k * pitch` and every lane predicate derived from them
// out of the loop - ~50 live vector registers and ~60 scalar register pairs of lane masks, which it then
// spills to vector-register lanes (v_writelane / v_readlane + hazard s_nops on every use: a quarter of
// the loop's vector instructions in round 1).  Recomputing costs one add or compare each.
__device__ __forceinline__ int ec_opaque(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ unsigned ec_opaque(unsigned x)
{
    asm volatile("" : "+v"(x));
    return x;
}

// The same for a wave-uniform value (scalar register): every scalar derived from it (row pitches times k, plane
// bases, table offsets) is then recomputed by two or three scalar instructions where it is used instead of being
// kept live over the whole time loop - ~120 of those did not fit the 102 scalar registers and were spilled.
__device__ __forceinline__ int ec_su(int x)
{
    x = __builtin_amdgcn_readfirstlane(x);          // folded away when x already sits in a scalar register
    asm volatile("" : "+s"(x));
    return x;
}
__device__ __forceinline__ unsigned ec_su(unsigned x) { return (unsigned)ec_su((int)x); }

// base + byte offset with the offset in a 32-bit vector register: selects the `global_* v_off, data, s[base]`
// addressing form (uniform 64-bit base in scalar registers) instead of 64-bit vector address arithmetic
template <class T>
__device__ __forceinline__ T *ec_at(T *base, unsigned byte_off)
{
    return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off);
}
template <class T>
__device__ __forceinline__ const T *ec_at(const T *base, unsigned byte_off)
{
    return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}

// All LDS reads of a group update are requested in one batch and pinned here: the values become opaque, so the
// compiler neither re-reads misaligned pairs from LDS in the middle of the arithmetic (it did: five dependent LDS
// round trips per update instead of one) nor moves the reads apart.
__device__ __forceinline__ void ec_pin(float4 &a, float4 &b, float4 &c, float4 &d)
{
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w),
                      "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w), "+v"(d.x), "+v"(d.y), "+v"(d.z), "+v"(d.w));
}
__device__ __forceinline__ void ec_pin(float4 &a, float4 &b)
{
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(b.w));
}
__device__ __forceinline__ void ec_pin(float2 &a, float2 &b, float2 &c, float2 &d)
{
    asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(b.x), "+v"(b.y), "+v"(c.x), "+v"(c.y), "+v"(d.x), "+v"(d.y));
}

// nothing of the prologue is in flight when the time loop starts: without this the compiler carries the
// prologue's global loads as "possibly pending" into the loop and waits on vmcnt before the first use of
// each such register in EVERY iteration - where the wait then drains the snapshot stores and publishes.
__device__ __forceinline__ void ec_drain_vmem() { __builtin_amdgcn_s_waitcnt(0x0f70); }   // vmcnt(0)
// A value requested long ago is declared arrived HERE (the wait the compiler puts in front of this is free where the
// caller knows the queue has drained): its later use no longer waits for whatever was requested in between.
__device__ __forceinline__ void ec_settle(float &x) { asm volatile("" : "+v"(x)); }

// Publishes stay in the XCD's L2 (workgroup-scope stores; mifwi::same_xcd in mifwi_common.h checks the placement that
// makes this correct).  AG = true (the kernels' last template argument): agent-scope stores, written through the
// fabric - correct on any placement; the host launches these variants after a failed placement check.
template <bool AG> struct EcScope { static constexpr int value = AG ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_WORKGROUP; };

// The granule hand-off of one slab, shared by the forward and the adjoint kernel.  `kind` selects one
// of the two exchanges of a step, `parity` the double buffer, `epoch` the tag a complete granule
// carries.  receive() hands every arrived value to dest(lds_offset, value); lds_offset already holds
// the plane of the field (B-field plane + one plane for the A field, see plane_a).
struct EcHandoff {
    unsigned long long *xw;              // this slab's slots [kind 0..1][parity 0..1][kEcRowFields*gp]
    unsigned xslot8;                     // bytes of one slot
    int gp, ng, NW, w, R, t;
    int rcv_lo[kEcGr];                   // LDS offset (B-field plane = 0, A-field plane = plane_a) of my k-th granule, <0: none
    unsigned rcv_off[kEcGr];             // its byte offset from (xw - 4 slots): the slab above sits there, the one below 8 slots on
    int *err;
    bool failed, no_wait;                // no_wait: timing ablation only (wrong results)
    int nap;                             // s_sleep units between poll passes (mifwi::poll_nap)

    __device__ __forceinline__ void init(unsigned long long *xbuf, int s, int NW_, int w_, int R_, int PL, int gp_,
                                         int ng_, int t_, int *err_, bool no_wait_, int nap_, int plane_a)
    {
        nap = nap_;
        gp = gp_; ng = ng_; NW = NW_; w = w_; R = R_; t = t_; err = err_; failed = false; no_wait = no_wait_;
        xslot8 = 8u * kEcRowFields * gp;
        xw = xbuf + ((long long)s * NW + w) * 4 * (kEcRowFields * gp);
#pragma unroll
        for (int k = 0; k < kEcGr; ++k) {
            const int e = t + k * kEcThreads;
            rcv_lo[k] = -1;
            rcv_off[k] = 4u * xslot8;               // no granule: reads the head of its own slot, value unused
            if (e < kEcRowFields * gp) {
                const int cq = e % gp, rf = e / gp;
                const int col = 4 * (cq % ng) + cq / ng;             // granule order is [k][group]
                const bool from_above = rf >= 3;                      // top halo <- the slab above's last rows
                if (cq < 4 * ng && !(from_above ? w == 0 : w == NW - 1)) {
                    rcv_lo[k] = ec_rf_lds_row(rf, R) * PL + 4 + col + (ec_rf_field(rf) ? plane_a : 0);
                    rcv_off[k] = 8u * e + (from_above ? 0u : 8u * xslot8);          // same index on both sides
                }
            }
        }
    }

    // Hand-off receive in two halves:
    //   request(): one pass of loads of ALL of this thread's granules, back to back (one memory round trip per
    //              pass, not one per granule), nothing waits;
    //   complete(): examine that pass; while a tag is missing, nap and sweep again (bounded: a time-out sets
    //              `failed`); then hand every value to dest(lds_offset, value).
    struct Pending {
        const unsigned long long *base;
        unsigned long long v[kEcGr];
        __device__ __forceinline__ void clear()
        {
            base = nullptr;
#pragma unroll
            for (int k = 0; k < kEcGr; ++k) v[k] = 0;
        }
    };
    __device__ __forceinline__ void sweep(Pending &q) const
    {
#pragma unroll
        for (int k = 0; k < kEcGr; ++k)       // lanes without a k-th granule read their own slot: no branches
            q.v[k] = __hip_atomic_load(ec_at(q.base, ec_opaque(rcv_off[k])), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void request(Pending &q, int kind, int parity) const
    {
        // (xw - 4 slots) + slot of (kind, parity): uniform base, per-lane byte offsets
        const unsigned xs8 = ec_su(xslot8);
        q.base = ec_at(xw, (unsigned)(kind * 2 + parity) * xs8) - 4 * (xs8 / 8);
        sweep(q);
    }
    template <class Dest>
    __device__ __forceinline__ void complete(Pending &q, unsigned epoch, Dest dest)
    {
        int lo[kEcGr];
        unsigned need[kEcGr];                  // all ones where this thread has a k-th granule
#pragma unroll
        for (int k = 0; k < kEcGr; ++k) { lo[k] = ec_opaque(rcv_lo[k]); need[k] = ~(unsigned)(lo[k] >> 31); }
        for (unsigned spins = 0;; ++spins) {
            unsigned bad = 0;
#pragma unroll
            for (int k = 0; k < kEcGr; ++k) bad |= ((unsigned)(q.v[k] >> 32) ^ epoch) & need[k];
            // a wave that waits this long for a neighbour says so at once (rare path, no state carried through the loop)
            if (spins == mifwi::kSlowPollPasses && (t & 63) == 0) atomicAdd(err + mifwi::kErrSlow, 1);
            if (bad == 0 || no_wait || failed) break;   // once failed: one pass per hand-off, garbage forward until the check
            if (spins > kEcMaxSpin ||
                ((spins & 255u) == 255u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                // publish at once: every other workgroup of the launch bails within 256 spins instead of running
                // into its own time-out, one hand-off after the other
                if (spins > kEcMaxSpin) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                failed = true;
                break;
            }
            mifwi::poll_nap(nap);
            sweep(q);
        }
#pragma unroll
        for (int k = 0; k < kEcGr; ++k)
            if (lo[k] >= 0) dest(lo[k], __uint_as_float((unsigned)q.v[k]));
        // Vector-memory operations of a wave complete in order, so with the last sweep landed nothing this wave
        // issued before it is in flight any more: vmcnt(0) costs nothing here, and it is the only way to tell the
        // compiler.  Without it the registers of the sweep count as "possibly pending" on some path through the
        // spin loop, and the first reuse of each one in the update that follows is guarded by an s_waitcnt vmcnt -
        // which by then waits for the snapshot traffic issued in between (adjoint: 1500 clocks per step).
        ec_drain_vmem();
    }
    template <class Dest>
    __device__ __forceinline__ void receive(int kind, unsigned epoch, int parity, Dest dest)
    {
        Pending q;
        request(q, kind, parity);
        complete(q, epoch, dest);
    }

    // publish the four cells of a boundary-row group (local row lrw, group gq): b = the field a forward
    // difference reads (vx, szz; E2, D2), a = the one a backward difference reads (vz, sxz; E3, D4)
    template <bool AG>
    __device__ __forceinline__ void publish(int lrw, int gq, int kind, unsigned epoch, int parity, const float4 &b,
                                            const float4 &a) const
    {
        unsigned long long *x = ec_at(xw, (unsigned)(kind * 2 + parity) * ec_su(xslot8));
        const unsigned long long tag = (unsigned long long)epoch << 32;
        const float bv[4] = {b.x, b.y, b.z, b.w}, av[4] = {a.x, a.y, a.z, a.w};
        const unsigned gp8 = 8u * ec_su((unsigned)gp), ng8 = 8u * ec_su((unsigned)ng), gq8 = 8u * gq;
        auto put = [&](int rf, const float (&v)[4]) {
            unsigned o = rf * gp8 + gq8;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                __hip_atomic_store(ec_at(x, o), tag | __float_as_uint(v[k]), __ATOMIC_RELAXED, EcScope<AG>::value);
                o = ec_opaque(o + ng8);                            // a chain of adds, not 24 hoisted constants
            }
        };
        const int wq = ec_su(w), Rq = ec_su(R);
        if (wq > 0) {
            if (lrw == 0) { put(0, av); put(1, bv); }
            if (lrw == 1) put(2, bv);
        }
        if (wq < ec_su(NW) - 1) {
            if (lrw == Rq - 2) put(3, av);
            if (lrw == Rq - 1) { put(4, av); put(5, bv); }
        }
    }
};

// per-thread state of one owned group of 4 cells (registers for the whole run)
// Which group thread t holds in slot q (v = t + q * kEcThreads).  The interior groups of the slab come first (rows
// 2 .. R-3, row-major), the boundary groups (rows 0, 1, R-2, R-1) start at the next wave boundary where that fits: a
// wave's slot then holds ONE class.  class 1 = interior, first slot: updated before the poll; 2 = boundary: after it;
// 3 = interior, second slot: updated LATE, together with the boundary rows.  With the plain row-major deal
// (v -> row v / ng) three of the eight waves of a 13-row slab of 100x300 made two interior passes before the poll while
// five waited, then idled while those five did the boundary rows, and two waves held both classes in one slot (a
// whole extra pass for 7 and 22 lanes): 2 I + B per half step on the critical path.  Now every wave makes one interior
// pass, and the second interior pass of three waves runs beside the boundary pass of the others: I + max(I, B).
#ifndef EC_LATE_INTERIOR
#define EC_LATE_INTERIOR 1
#endif
#ifndef EA_PUBLISH_FIRST
#define EA_PUBLISH_FIRST 1
#endif
#ifndef EA_MATS_AHEAD_A
#define EA_MATS_AHEAD_A 0          // measured on 100x300: A ahead 8.81 us, C ahead 8.41, both 8.50 (13 VGPRs spilled), none 8.58
#endif
#ifndef EA_MATS_AHEAD_C
#define EA_MATS_AHEAD_C 1
#endif
struct EcSlot { int lrw, g, cls; };
__host__ __device__ __forceinline__ EcSlot ec_slot(int v, int R, int ng, int slots)       // slots = NG * kEcThreads >= R * ng
{
    EcSlot s;
    s.lrw = 0; s.g = 0; s.cls = 0;
#if EC_LATE_INTERIOR
    const int nint = (R > 4 ? R - 4 : 0) * ng, nb = R * ng - nint;
    int b0 = (nint + 63) & ~63;
    if (b0 + nb > slots) b0 = nint;                          // no room for the pad: classes share a wave's slot
    if (v < nint) {
        const int ir = v / ng;
        s.lrw = 2 + ir; s.g = v - ir * ng; s.cls = v < kEcThreads ? 1 : 3;
    } else if (v >= b0 && v < b0 + nb) {
        const int bi = v - b0, br = bi / ng;
        s.g = bi - br * ng;
        s.lrw = (R > 4 && br >= 2) ? R - 4 + br : br;
        s.cls = 2;
    }
#else
    if (v < R * ng) {
        s.lrw = v / ng; s.g = v - s.lrw * ng;
        s.cls = (s.lrw >= 2 && s.lrw < R - 2) ? 1 : 2;
    }
#endif
    return s;
}

// The x-stencils of a group of four cells reach two cells into the groups left and right of it.  Those cells sit in the
// registers of the neighbouring LANES (the deal is row-major: consecutive lanes hold consecutive groups), which have just
// read them from LDS as part of their own group: a DPP wave shift fetches them instead of two 8-byte LDS reads at
// misaligned offsets per field (two lanes per bank by construction; together with the row breaks of the 16-byte reads the
// LDS pipe spent more cycles on bank conflicts than on accesses: SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = 1.6 forward,
// 1.3 adjoint).  A lane at the end of a grid row EXPORTS zeros (what the planes' side pads hold) to the lane that starts
// the next row; a lane without an active lane of its class next to it (wave ends, the ends of a class's lanes) keeps the
// value of ONE LDS read per field - the other lanes of that read all fetch one address, a conflict-free broadcast - as
// the DPP move's `old` operand.  Straight-line code: a branch around it costs more than the reads it saves (measured), so
// the kernels come in two variants (template XH) and the host launches XH = true only for deals in which no lane needs
// two reads and no two classes that run in one pass meet inside a wave (ec_xhalo_deal_ok).
// ec_nb: kNbL / kNbR = the lane below / above holds a group of the same class; kNbFirst / kNbLast = first / last group of
// a grid row.
constexpr int kNbL = 1, kNbR = 2, kNbFirst = 4, kNbLast = 8;
__host__ __device__ __forceinline__ int ec_nb(int v, int R, int ng, int slots)
{
    const EcSlot s = ec_slot(v, R, ng, slots);
    if (s.cls == 0) return 0;
    const int lane = v & 63;                          // (kEcThreads is a multiple of 64)
    const bool lv = lane != 0 && ec_slot(v - 1, R, ng, slots).cls == s.cls;
    const bool rv = lane != 63 && ec_slot(v + 1, R, ng, slots).cls == s.cls;
    return (lv ? kNbL : 0) | (rv ? kNbR : 0) | (s.g == 0 ? kNbFirst : 0) | (s.g == ng - 1 ? kNbLast : 0);
}
// host: may the XH variants run the deal of a slab of R rows?
inline bool ec_xhalo_deal_ok(int R, int ng, int slots)
{
    for (int v = 0; v < slots; ++v) {
        const EcSlot s = ec_slot(v, R, ng, slots);
        if (s.cls == 0) continue;
        const int nb = ec_nb(v, R, ng, slots), lane = v & 63;
        if (!(nb & (kNbL | kNbR))) return false;                               // a lane that would need two reads
        // boundary (2) and late-interior (3) groups run in ONE pass of the adjoint's phase B: side by side in a wave the
        // shift would hand a lane the cells of another row
        const int lc = lane != 0 ? ec_slot(v - 1, R, ng, slots).cls : 0;
        if (lc >= 2 && s.cls >= 2 && lc != s.cls) return false;
    }
    return true;
}
__device__ __forceinline__ float ec_lane_below(float old, float x)          // x of lane - 1 (wave_shr:1); `old` without one
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float ec_lane_above(float old, float x)          // x of lane + 1 (wave_shl:1)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(x), 0x130, 0xf, 0xf, false));
}
// the two cells left (L) and right (R) of a group (own four cells c, plane row at q); dummy = any 8-byte aligned LDS address
template <bool XH>
__device__ __forceinline__ void ec_xhalo(const float *q, const float4 &c, const int nb, const float *dummy, float2 &L, float2 &R)
{
    if (!XH) {
        L = ld2(q - 2); R = ld2(q + 4);
        return;
    }
    const float2 rd = ld2(!(nb & kNbL) ? q - 2 : !(nb & kNbR) ? q + 4 : dummy);
    const float ez = (nb & kNbLast) ? 0.f : c.z, ew = (nb & kNbLast) ? 0.f : c.w;
    const float ex = (nb & kNbFirst) ? 0.f : c.x, ey = (nb & kNbFirst) ? 0.f : c.y;
    L.x = ec_lane_below(rd.x, ez); L.y = ec_lane_below(rd.y, ew);
    R.x = ec_lane_above(rd.x, ex); R.y = ec_lane_above(rd.y, ey);
}

struct EcGroup {
    int cls;                                      // 0: none, 1: interior rows of the slab, 2: boundary rows (stencils reach the halo)
    int nb;                                       // ec_nb: how the x-neighbours' cells reach this lane
    int g, j;                                     // group in the row, grid row
    int lo;                                       // LDS float offset of the group inside a field plane
    unsigned gcb;                                 // byte offset of the group inside a [nz][gp] plane (snapshots, materials)
    int xtab, ztab;                               // LDS float offsets of its C-PML table entries (x strip / z strip), or -1
    float4 mL, mM, mMu, mBx, mBz;                 // materials
    float4 s1, s2, s3, s4, s5, s6, s7, s8;        // C-PML memory variables
};

// LDS planes of the forward kernel: [vx | vz | szz | sxz | sxx], so that for both hand-offs the field a
// backward difference reads sits one plane after the one a forward difference reads (EcHandoff::rcv_lo)
__device__ __forceinline__ int ec_plane(int f) { return f == F_VX ? 0 : f == F_VZ ? 1 : f == F_SZZ ? 2 : f == F_SXZ ? 3 : 4; }

// C-PML tables in LDS, one entry = the six profiles of a cell side by side (immediate offsets, no pitch
// arithmetic): x table [group][6][4 cells], z table [row][8] (6 used)
__device__ __forceinline__ void ec_stage_tables(float *lpx, float *lpz, const float *px, const float *pz, int gp, int nz,
                                                int r0, int R, int t)
{
    for (int e = t; e < 6 * gp; e += kEcThreads) {
        const int k = e / gp, col = e - k * gp;
        lpx[((col >> 2) * 6 + k) * 4 + (col & 3)] = px[e];
    }
    for (int e = t; e < 8 * R; e += kEcThreads) {
        const int k = e & 7, r = e >> 3;
        lpz[e] = k < 6 ? pz[k * nz + r0 + r] : 0.f;
    }
}
struct EcTabX { float4 v[6]; };
struct EcTabZ { float v[8]; };
__device__ __forceinline__ EcTabX ec_tab_x(const float *lpx, int xtab)
{
    EcTabX r;
#pragma unroll
    for (int k = 0; k < 6; ++k) r.v[k] = ld4(lpx + xtab + 4 * k);
    return r;
}
__device__ __forceinline__ EcTabZ ec_tab_z(const float *lpz, int ztab)
{
    const float4 a = ld4(lpz + ztab);
    const float2 b = ld2(lpz + ztab + 4);
    EcTabZ r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = 0.f; r.v[7] = 0.f;
    return r;
}

struct EcCtx {
    float *Lf[5];
    const float *lpx, *lpz;
    int PL, fsurf;
    FdK K;
};

// V update (reads stresses from LDS, writes the group's velocities in place)
template <bool EDGE, bool XH>   // EDGE: the group may sit on grid rows 0/1 (free-surface mirroring); interior rows never do
__device__ __forceinline__ void ec_update_v(EcGroup &G, const EcCtx &c, const int lo, const int jq, float4 &S4, float4 &S5,
                                            float4 &o0, float4 &o1)
{
    const int PL = ec_su(c.PL);
    const FdK K = c.K;
    const float *sxx = c.Lf[F_SXX] + lo, *szz = c.Lf[F_SZZ] + lo, *sxz = c.Lf[F_SXZ] + lo;
    const int nb = XH ? ec_opaque(G.nb) : 0;
    float4 cxx = ld4(sxx);
    float4 a2 = ld4(sxz);
    float2 Lxx, Rxx, Lxz, Rxz;
    float4 a0 = ld4(sxz - 2 * PL), a1 = ld4(sxz - PL);
    float4 a3 = ld4(sxz + PL);
    float4 b0 = ld4(szz - PL);
    float4 b1 = ld4(szz), b2 = ld4(szz + PL), b3 = ld4(szz + 2 * PL);
    ec_xhalo<XH>(sxx, cxx, nb, c.Lf[0], Lxx, Rxx);
    ec_xhalo<XH>(sxz, a2, nb, c.Lf[0], Lxz, Rxz);
#ifdef EC_PIN
    ec_pin(Lxx, Rxx, Lxz, Rxz);
    ec_pin(cxx, a2, a0, a1);
    ec_pin(a3, b0, b1, b2);
    ec_pin(b3, b3);
#endif
    if (EDGE && c.fsurf && jq < 2) {
        if (jq == 0) {
            a1 = make_float4(-a2.x, -a2.y, -a2.z, -a2.w);
            a0 = make_float4(-a3.x, -a3.y, -a3.z, -a3.w);
            b0 = make_float4(-b2.x, -b2.y, -b2.z, -b2.w);
        } else {
            a0 = make_float4(-a1.x, -a1.y, -a1.z, -a1.w);
        }
    }
    const float xx[8] = {Lxx.x, Lxx.y, cxx.x, cxx.y, cxx.z, cxx.w, Rxx.x, Rxx.y};
    const float xz[8] = {Lxz.x, Lxz.y, a2.x, a2.y, a2.z, a2.w, Rxz.x, Rxz.y};
    float d1[4], d2[4], d3[4], d4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        d1[k] = dfw(K, xx[k + 1], xx[k + 2], xx[k + 3], xx[k + 4]);
        d2[k] = dbw(K, comp(a0, k), comp(a1, k), comp(a2, k), comp(a3, k));
        d3[k] = dbw(K, xz[k], xz[k + 1], xz[k + 2], xz[k + 3]);
        d4[k] = dfw(K, comp(b0, k), comp(b1, k), comp(b2, k), comp(b3, k));
    }
    const int xtab = ec_opaque(G.xtab), ztab = ec_opaque(G.ztab);
    if (xtab >= 0) {
        const EcTabX T = ec_tab_x(c.lpx, xtab);
        float t1[4] = {G.s1.x, G.s1.y, G.s1.z, G.s1.w}, t3[4] = {G.s3.x, G.s3.y, G.s3.z, G.s3.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d1[k] = pml(t1[k], comp(T.v[PAH], k), comp(T.v[PBH], k), comp(T.v[PKH], k), d1[k]);
            d3[k] = pml(t3[k], comp(T.v[PA], k), comp(T.v[PB], k), comp(T.v[PK], k), d3[k]);
        }
        G.s1 = make_float4(t1[0], t1[1], t1[2], t1[3]); G.s3 = make_float4(t3[0], t3[1], t3[2], t3[3]);
    }
    if (ztab >= 0) {
        const EcTabZ Z = ec_tab_z(c.lpz, ztab);
        float t2[4] = {G.s2.x, G.s2.y, G.s2.z, G.s2.w}, t4[4] = {G.s4.x, G.s4.y, G.s4.z, G.s4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d2[k] = pml(t2[k], Z.v[PA], Z.v[PB], Z.v[PK], d2[k]);
            d4[k] = pml(t4[k], Z.v[PAH], Z.v[PBH], Z.v[PKH], d4[k]);
        }
        G.s2 = make_float4(t2[0], t2[1], t2[2], t2[3]); G.s4 = make_float4(t4[0], t4[1], t4[2], t4[3]);
    }
    const float4 vxo = ld4(c.Lf[F_VX] + lo), vzo = ld4(c.Lf[F_VZ] + lo);
    S4 = make_float4(d1[0] + d2[0], d1[1] + d2[1], d1[2] + d2[2], d1[3] + d2[3]);
    S5 = make_float4(d3[0] + d4[0], d3[1] + d4[1], d3[2] + d4[2], d3[3] + d4[3]);
    o0 = make_float4(fmaf(G.mBx.x, S4.x, vxo.x), fmaf(G.mBx.y, S4.y, vxo.y), fmaf(G.mBx.z, S4.z, vxo.z),
                     fmaf(G.mBx.w, S4.w, vxo.w));
    o1 = make_float4(fmaf(G.mBz.x, S5.x, vzo.x), fmaf(G.mBz.y, S5.y, vzo.y), fmaf(G.mBz.z, S5.z, vzo.z),
                     fmaf(G.mBz.w, S5.w, vzo.w));
    st4(c.Lf[F_VX] + lo, o0);
    st4(c.Lf[F_VZ] + lo, o1);
}

// S update (reads velocities from LDS, writes the group's stresses in place); `amp` = source term
// of this step for the group's cells (0 where there is none)
template <bool EDGE, bool XH>
__device__ __forceinline__ void ec_update_s(EcGroup &G, const EcCtx &c, const int lo, const int jq, const float4 &amp, float4 &S1,
                                            float4 &S2, float4 &S3, float4 &o0, float4 &o1)
{
    const int PL = ec_su(c.PL);
    const FdK K = c.K;
    const float *vx = c.Lf[F_VX] + lo, *vz = c.Lf[F_VZ] + lo;
    const int nb = XH ? ec_opaque(G.nb) : 0;
    float4 b1 = ld4(vx);
    float4 a2 = ld4(vz);
    float2 Lvx, Rvx, Lvz, Rvz;
    float4 a0 = ld4(vz - 2 * PL), a1 = ld4(vz - PL), a3 = ld4(vz + PL);
    float4 b0 = ld4(vx - PL), b2 = ld4(vx + PL), b3 = ld4(vx + 2 * PL);
    ec_xhalo<XH>(vx, b1, nb, c.Lf[0], Lvx, Rvx);
    ec_xhalo<XH>(vz, a2, nb, c.Lf[0], Lvz, Rvz);
#ifdef EC_PIN
    ec_pin(Lvx, Rvx, Lvz, Rvz);
    ec_pin(b1, a2, a0, a1);
    ec_pin(a3, b0, b2, b3);
#endif
    const float xv[8] = {Lvx.x, Lvx.y, b1.x, b1.y, b1.z, b1.w, Rvx.x, Rvx.y};
    const float zv[8] = {Lvz.x, Lvz.y, a2.x, a2.y, a2.z, a2.w, Rvz.x, Rvz.y};
    float e1[4], e2[4], e3[4], e4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        e1[k] = dbw(K, xv[k], xv[k + 1], xv[k + 2], xv[k + 3]);
        e2[k] = dbw(K, comp(a0, k), comp(a1, k), comp(a2, k), comp(a3, k));
        e3[k] = dfw(K, comp(b0, k), comp(b1, k), comp(b2, k), comp(b3, k));
        e4[k] = dfw(K, zv[k + 1], zv[k + 2], zv[k + 3], zv[k + 4]);
    }
    const int xtab = ec_opaque(G.xtab), ztab = ec_opaque(G.ztab);
    if (xtab >= 0) {
        const EcTabX T = ec_tab_x(c.lpx, xtab);
        float t5[4] = {G.s5.x, G.s5.y, G.s5.z, G.s5.w}, t8[4] = {G.s8.x, G.s8.y, G.s8.z, G.s8.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            e1[k] = pml(t5[k], comp(T.v[PA], k), comp(T.v[PB], k), comp(T.v[PK], k), e1[k]);
            e4[k] = pml(t8[k], comp(T.v[PAH], k), comp(T.v[PBH], k), comp(T.v[PKH], k), e4[k]);
        }
        G.s5 = make_float4(t5[0], t5[1], t5[2], t5[3]); G.s8 = make_float4(t8[0], t8[1], t8[2], t8[3]);
    }
    if (ztab >= 0) {
        const EcTabZ Z = ec_tab_z(c.lpz, ztab);
        float t6[4] = {G.s6.x, G.s6.y, G.s6.z, G.s6.w}, t7[4] = {G.s7.x, G.s7.y, G.s7.z, G.s7.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            e2[k] = pml(t6[k], Z.v[PA], Z.v[PB], Z.v[PK], e2[k]);
            e3[k] = pml(t7[k], Z.v[PAH], Z.v[PBH], Z.v[PKH], e3[k]);
        }
        G.s6 = make_float4(t6[0], t6[1], t6[2], t6[3]); G.s7 = make_float4(t7[0], t7[1], t7[2], t7[3]);
    }
    const float4 oxx = ld4(c.Lf[F_SXX] + lo), ozz = ld4(c.Lf[F_SZZ] + lo), oxz = ld4(c.Lf[F_SXZ] + lo);
    float rxx[4], rzz[4], rxz[4], s3v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s3v[k] = e3[k] + e4[k];
        rxx[k] = fmaf(comp(G.mM, k), e1[k], fmaf(comp(G.mL, k), e2[k], comp(oxx, k)));
        rzz[k] = fmaf(comp(G.mL, k), e1[k], fmaf(comp(G.mM, k), e2[k], comp(ozz, k)));
        rxz[k] = fmaf(comp(G.mMu, k), s3v[k], comp(oxz, k));
        rxx[k] += comp(amp, k); rzz[k] += comp(amp, k);       // source term of the cell (0 without one)
    }
    if (EDGE && c.fsurf && jq == 0) { rzz[0] = rzz[1] = rzz[2] = rzz[3] = 0.f; }
    S1 = make_float4(e1[0], e1[1], e1[2], e1[3]); S2 = make_float4(e2[0], e2[1], e2[2], e2[3]);
    S3 = make_float4(s3v[0], s3v[1], s3v[2], s3v[3]);
    o0 = make_float4(rzz[0], rzz[1], rzz[2], rzz[3]);            // published: szz, sxz
    o1 = make_float4(rxz[0], rxz[1], rxz[2], rxz[3]);
    st4(c.Lf[F_SXX] + lo, make_float4(rxx[0], rxx[1], rxx[2], rxx[3]));
    st4(c.Lf[F_SZZ] + lo, o0);
    st4(c.Lf[F_SXZ] + lo, o1);
}

template <bool SAVE, int NG, bool AG, bool XH = false>
__global__ __launch_bounds__(kEcThreads) void el_cluster_fwd(const EcParams p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int L = (int)blockIdx.x;
    const int xcd = L & 7, kq = L >> 3;
    const int w = kq % p.NW, s = p.shot0 + xcd + 8 * (kq / p.NW);
    if (s >= p.shot1) return;
    // ablation builds only: slab 1 of the first shot never shows up (a workgroup that was not resident in time)
    if ((kDbg(p) & 64) && w == 1 && s == p.shot0) return;
    const int t = (int)threadIdx.x;
    if (!AG && !mifwi::same_xcd(p.xcc_tab, s, p.NW, w, t, p.err, kEcMaxSpin, kDbg(p) & 128)) return;
    int r0, R;
    ec_slab_rows(p.nz, p.NW, w, r0, R);
    const int PL = p.PL, LR = R + 4;
    const int fsz = LR * PL;
    EcCtx c;
    for (int k = 0; k < 5; ++k) c.Lf[k] = lds + ec_plane(k) * fsz;
    float *lpx = lds + 5 * fsz, *lpz = lpx + 6 * p.gp;        // C-PML tables (ec_stage_tables)
    c.lpx = lpx; c.lpz = lpz; c.PL = PL; c.fsurf = p.fsurf; c.K = p.K;
    const unsigned ncell = (unsigned)p.nz * p.gp;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long xplane = (long long)p.nz * p.wx, zplane = 2LL * p.W * p.gp;
    // strip slots of a group in the global C-PML state (or -1)
    auto strip_x = [&](int g) { const int c0 = 4 * g; return p.W <= 0 ? -1 : c0 < p.wl ? c0 : c0 >= p.xr0 ? p.wl + (c0 - p.xr0) : -1; };
    auto strip_z = [&](int j) { return p.W <= 0 ? -1 : j < p.W ? j : j >= p.nz - p.W ? j - (p.nz - 2 * p.W) : -1; };

    // ---- this thread's groups ---------------------------------------------------------------------
    EcGroup G[NG];
    bool slow = p.nrec > kEcThreads;
    int tsq = -1;                        // group slot holding this thread's source cell (fast path: at most one), or -1
    unsigned tsrc4 = 0;                  // byte offset of that source inside a [nsrc] row of f
    unsigned tm0 = 0, tm1 = 0, tm2 = 0, tm3 = 0;   // all ones on the source's cell of the group (exact +0 elsewhere)
    float tw = 0.f, tnext = 0.f;         // its weight; f of the next step for that source, fetched one step ahead
    int src_mask = 0;                    // slots with sources (slow path)
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        EcGroup &g = G[q];
        const EcSlot sl = ec_slot(t + q * kEcThreads, R, p.ng, NG * kEcThreads);
        const bool own = sl.cls != 0;
        const int lrw = sl.lrw;
        g.g = sl.g;
        g.j = r0 + lrw;
        g.cls = sl.cls;                      // 1, 3: stencils stay inside the own rows
        g.nb = XH ? ec_nb(t + q * kEcThreads, R, p.ng, NG * kEcThreads) : 0;
        g.lo = (lrw + 2) * PL + 4 + 4 * g.g;
        const unsigned gcc = (unsigned)g.j * p.gp + 4 * g.g;
        g.gcb = 4u * gcc;
        g.mL = g.mM = g.mMu = g.mBx = g.mBz = zero4;
        g.s1 = g.s2 = g.s3 = g.s4 = g.s5 = g.s6 = g.s7 = g.s8 = zero4;
        g.xtab = -1; g.ztab = -1;
        int xs_off = -1, zs = -1;
        if (own) {
            g.mL = ld4(p.mat + M_L * ncell + gcc); g.mM = ld4(p.mat + M_M * ncell + gcc);
            g.mMu = ld4(p.mat + M_MU * ncell + gcc);
            g.mBx = ld4(p.mat + M_BX * ncell + gcc); g.mBz = ld4(p.mat + M_BZ * ncell + gcc);
            xs_off = strip_x(g.g); zs = strip_z(g.j);
        }
        if (xs_off >= 0) {
            g.xtab = 24 * g.g;
            const float *q = p.psix + (long long)s * p.psix_shot + (long long)g.j * p.wx + xs_off;
            g.s1 = ld4(q); g.s3 = ld4(q + xplane); g.s5 = ld4(q + 2 * xplane); g.s8 = ld4(q + 3 * xplane);
        }
        if (zs >= 0) {
            g.ztab = 8 * lrw;
            const float *q = p.psiz + (long long)s * p.psiz_shot + (long long)zs * p.gp + 4 * g.g;
            g.s2 = ld4(q); g.s4 = ld4(q + zplane); g.s6 = ld4(q + 2 * zplane); g.s7 = ld4(q + 3 * zplane);
        }
        // at most one source tap per thread (otherwise: slow path, rescan per step)
        for (int e = 0; e < p.nsrc; ++e) {
            const int cell = p.src_cell[(long long)s * p.nsrc + e];
            if (cell < 0) continue;
            const int i0 = cell / p.nx, i1 = cell - i0 * p.nx;
            if (own && i0 == g.j && (i1 >> 2) == g.g) {
                if (tsq >= 0) slow = true;
                tw = p.src_w[(long long)s * p.nsrc + e];
                tsq = q; tsrc4 = 4u * e;
                tm0 = (i1 & 3) == 0 ? ~0u : 0u; tm1 = (i1 & 3) == 1 ? ~0u : 0u;
                tm2 = (i1 & 3) == 2 ? ~0u : 0u; tm3 = (i1 & 3) == 3 ? ~0u : 0u;
                src_mask |= 1 << q;
            }
        }
    }
    // one receiver per thread (otherwise: slow path)
    int smp_lo = -1;                     // LDS offset of my receiver (-2: inactive tap, slab 0 writes zeros)
    float smp_w = 0.f;
    if (p.rec_vx != nullptr && t < p.nrec) {
        const int cell = p.rec_cell[(long long)s * p.nrec + t];
        if (cell >= 0) {
            const int i0 = cell / p.nx, i1 = cell - i0 * p.nx;
            if (i0 >= r0 && i0 < r0 + R) { smp_lo = (i0 - r0 + 2) * PL + 4 + i1; smp_w = p.rec_w[(long long)s * p.nrec + t]; }
        } else if (w == 0) {
            smp_lo = -2;
        }
    }
    slow = __syncthreads_or(slow ? 1 : 0) != 0;
    const float *f_shot = p.f + (long long)s * p.nsrc;                 // f of step n: f_shot + n * f_step
    const long long f_step = (long long)p.nshot * p.nsrc;
    if (!slow && tsq >= 0 && p.n_first < p.n_last) tnext = *ec_at(f_shot + p.n_first * f_step, tsrc4);

    // ---- stage tables and the slab (+2 halo rows, + halo groups) of all five fields ----------------
    ec_stage_tables(lpx, lpz, p.px, p.pz, p.gp, p.nz, r0, R, t);
    {
        const float *gf = p.fields + (long long)s * p.shot_stride;
        const int ngl = PL / 4;
        for (int e = t; e < LR * ngl; e += kEcThreads) {
            const int lr = e / ngl, lg = e - lr * ngl;
            const int jj = r0 - 2 + lr, gg = lg - 1;
            const bool ok = jj >= 0 && jj < p.nz && gg >= 0 && gg < p.ng;
            const long long o = (long long)(jj + 2) * p.pitch + 4 + 4 * gg;
#pragma unroll
            for (int k = 0; k < 5; ++k) st4(c.Lf[k] + lr * PL + 4 * lg, ok ? ld4(gf + k * p.field_stride + o) : zero4);
        }
    }
    __syncthreads();

    // ---- halo hand-off: kind 0 = velocities after V, kind 1 = stresses after S ------------------
    EcHandoff X;
    X.init(p.xbuf, s, p.NW, w, R, PL, p.gp, p.ng, t, p.err, (kDbg(p) & 4) != 0, p.nap, fsz);
    const bool do_x = p.NW > 1 && !(kDbg(p) & 1);
    EcHandoff::Pending P;
    P.clear();
    auto complete = [&](int kind, unsigned epoch) {
        // planes [vx | vz] and [szz | sxz]: the offset already selects the second one for the A field
        float *base = kind == 0 ? c.Lf[F_VX] : c.Lf[F_SZZ];
        X.complete(P, epoch, [&](int off, float v) { base[off] = v; });
    };
    // source term of step n for the cells of a group
    auto source_amp = [&](const EcGroup &g, int q, int n) -> float4 {
        float4 a = zero4;
        if (!slow) {
            if (ec_opaque(tsq) == q) {
                // f of this step arrived during the last one; request the next (a global load in the update
                // itself would put a memory round trip on the workgroup's critical path every step)
                const unsigned amp = __float_as_uint(tw * tnext);
                a = make_float4(__uint_as_float(amp & tm0), __uint_as_float(amp & tm1), __uint_as_float(amp & tm2),
                                __uint_as_float(amp & tm3));
                if (n + 1 < p.n_last) tnext = *ec_at(f_shot + (n + 1) * f_step, ec_opaque(tsrc4));
            }
        } else if ((src_mask >> q) & 1) {   // several sources in this thread's cells: rescan the list every step
            float av[4] = {0.f, 0.f, 0.f, 0.f};
            for (int e = 0; e < p.nsrc; ++e) {
                const int cell = p.src_cell[(long long)s * p.nsrc + e];
                if (cell < 0) continue;
                const int i0 = cell / p.nx, i1 = cell - i0 * p.nx;
                if (i0 != g.j || (i1 >> 2) != g.g) continue;
                const float v = p.src_w[(long long)s * p.nsrc + e] * p.f[((long long)n * p.nshot + s) * p.nsrc + e];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k == (i1 & 3)) av[k] += v;
            }
            a = make_float4(av[0], av[1], av[2], av[3]);
        }
        return a;
    };

    const int nsteps = p.n_last - p.n_first;
    // Snapshot terms are stored right after they are produced; the interior update that follows gives
    // them time to retire before the next poll is issued (vector memory operations retire in order).
    float *S_shot = SAVE ? p.S + (long long)s * 5 * ncell : nullptr;   // step n: S_shot + (n - s_first) * s_step
    auto do_v = [&](EcGroup &g, int n, int it, bool edge) {
        float4 S4, S5, o0, o1;
        const int lo = ec_opaque(g.lo);
        if (edge) {
            const int jq = ec_opaque(g.j);
            ec_update_v<true, XH>(g, c, lo, jq, S4, S5, o0, o1);
            if (do_x && !(kDbg(p) & 16)) X.template publish<AG>(jq - r0, ec_opaque(g.g), 0, (unsigned)(2 * it + 1), it & 1, o0, o1);
        } else {
            ec_update_v<false, XH>(g, c, lo, 2, S4, S5, o0, o1);
        }
        if (SAVE && !(kDbg(p) & 2)) {
            float *Sn = S_shot + (long long)(n - p.s_first) * p.s_step;
            const unsigned gcb = ec_opaque(g.gcb), nc = ec_su(ncell);
            mifwi::stnt4(ec_at(Sn + 3 * (long long)nc, gcb), S4); mifwi::stnt4(ec_at(Sn + 4 * (long long)nc, gcb), S5);
        }
    };
    auto do_s = [&](EcGroup &g, int q, int n, int it, bool edge) {
        float4 S1, S2, S3, o0, o1;
        const int lo = ec_opaque(g.lo);
        const float4 amp = source_amp(g, q, n);
        if (edge) {
            const int jq = ec_opaque(g.j);
            ec_update_s<true, XH>(g, c, lo, jq, amp, S1, S2, S3, o0, o1);
            if (do_x && !(kDbg(p) & 16)) X.template publish<AG>(jq - r0, ec_opaque(g.g), 1, (unsigned)(2 * it + 2), it & 1, o0, o1);
        } else {
            ec_update_s<false, XH>(g, c, lo, 2, amp, S1, S2, S3, o0, o1);
        }
        if (SAVE && !(kDbg(p) & 2)) {
            float *Sn = S_shot + (long long)(n - p.s_first) * p.s_step;
            const unsigned gcb = ec_opaque(g.gcb), nc = ec_su(ncell);
            mifwi::stnt4(ec_at(Sn, gcb), S1); mifwi::stnt4(ec_at(Sn + (long long)nc, gcb), S2);
            mifwi::stnt4(ec_at(Sn + 2 * (long long)nc, gcb), S3);
        }
    };
#ifdef MIFWI_ABLATIONS
    const bool tr_on = w == (((kDbg(p) >> 8) & 15) ? ((kDbg(p) >> 8) & 15) - 1 : p.NW / 2) && s == p.shot0;   // dbg bits 8-11: traced slab + 1
#endif
    ec_drain_vmem();
    for (int it = 0; it < nsteps; ++it) {
        const int n = p.n_first + it;
        EC_LAGGARD();
        EC_STAMP(0);
        // ---- V: interior rows first, then receive the stress halo, then the boundary rows --------
        const bool poll_s = do_x && it > 0 && !(kDbg(p) & 32);
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            if ((!EC_LATE_INTERIOR || q == 0) && ec_opaque(G[q].cls) == 1) do_v(G[q], n, it, false);
            __builtin_amdgcn_sched_barrier(0);             // one group at a time: bounds the register peak
        }
        EC_STAMP(1);
        if (poll_s) {
            X.request(P, 1, (it - 1) & 1);
            complete(1, (unsigned)(2 * it));
        }
        EC_STAMP(2);
        __syncthreads();                                   // A: stress halo rows are in LDS
        EC_STAMP(3);
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const int cls = ec_opaque(G[q].cls);
            if (cls == 2) do_v(G[q], n, it, true);
            else if (q > 0 && cls == 3) do_v(G[q], n, it, false);          // late interior (ec_slot)
            __builtin_amdgcn_sched_barrier(0);
        }
        EC_STAMP(4);
        __syncthreads();                                   // B: all velocities of the slab are in LDS
        EC_STAMP(5);
        // ---- S: interior rows, receive the velocity halo, boundary rows -----------------------------
        const bool poll_v = do_x && !(kDbg(p) & 32);
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            if ((!EC_LATE_INTERIOR || q == 0) && ec_opaque(G[q].cls) == 1) do_s(G[q], q, n, it, false);
            __builtin_amdgcn_sched_barrier(0);
        }
        EC_STAMP(6);
        if (poll_v) {
            X.request(P, 0, it & 1);
            complete(0, (unsigned)(2 * it + 1));
        }
        EC_STAMP(7);
        // ---- receivers sample the new velocities (stores after the poll) ----------------------------
        if (p.rec_vx != nullptr && !(kDbg(p) & 8)) {
            if (!slow) {
                const long long ro = ((long long)n * p.nshot + s) * p.nrec;
                const int sl = ec_opaque(smp_lo);
                if (sl >= 0) {
                    *ec_at(p.rec_vx + ro, 4u * t) = fmaf(smp_w, c.Lf[F_VX][sl], 0.f);
                    *ec_at(p.rec_vz + ro, 4u * t) = fmaf(smp_w, c.Lf[F_VZ][sl], 0.f);
                } else if (sl == -2) {
                    *ec_at(p.rec_vx + ro, 4u * t) = 0.f; *ec_at(p.rec_vz + ro, 4u * t) = 0.f;
                }
            } else {
                for (int e = t; e < p.nrec; e += kEcThreads) {
                    const long long ro = ((long long)n * p.nshot + s) * p.nrec + e;
                    const int cell = p.rec_cell[(long long)s * p.nrec + e];
                    if (cell < 0) { if (w == 0) { p.rec_vx[ro] = 0.f; p.rec_vz[ro] = 0.f; } continue; }
                    const int i0 = cell / p.nx, i1 = cell - i0 * p.nx;
                    if (i0 >= r0 && i0 < r0 + R) {
                        const float ww = p.rec_w[(long long)s * p.nrec + e];
                        const int o = (i0 - r0 + 2) * PL + 4 + i1;
                        p.rec_vx[ro] = fmaf(ww, c.Lf[F_VX][o], 0.f);
                        p.rec_vz[ro] = fmaf(ww, c.Lf[F_VZ][o], 0.f);
                    }
                }
            }
        }
        EC_STAMP(8);
        __syncthreads();                                   // C: velocity halo rows are in LDS
        EC_STAMP(9);
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const int cls = ec_opaque(G[q].cls);
            if (cls == 2) do_s(G[q], q, n, it, true);
            else if (q > 0 && cls == 3) do_s(G[q], q, n, it, false);
            __builtin_amdgcn_sched_barrier(0);
        }
        EC_STAMP(10);
        if ((it & 31) == 31 || it == nsteps - 1) {
            if (__syncthreads_or(X.failed ? 1 : 0)) {        // D (+ collective time-out check)
                if (t == 0) __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        } else {
            __syncthreads();                               // D: all stresses of the slab are in LDS
        }
        EC_STAMP(11);
    }

    // ---- the own rows of the five fields and the memory variables go back to the global state -----
    float *gf = p.fields + (long long)s * p.shot_stride;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const EcGroup &g = G[q];
        if (g.cls == 0) continue;
        const long long o = (long long)(g.j + 2) * p.pitch + 4 + 4 * g.g;
#pragma unroll
        for (int k = 0; k < 5; ++k) st4(gf + k * p.field_stride + o, ld4(c.Lf[k] + g.lo));
        const int xs_off = strip_x(g.g), zs = strip_z(g.j);
        if (xs_off >= 0) {
            float *q2 = p.psix + (long long)s * p.psix_shot + (long long)g.j * p.wx + xs_off;
            st4(q2, g.s1); st4(q2 + xplane, g.s3); st4(q2 + 2 * xplane, g.s5); st4(q2 + 3 * xplane, g.s8);
        }
        if (zs >= 0) {
            float *q2 = p.psiz + (long long)s * p.psiz_shot + (long long)zs * p.gp + 4 * g.g;
            st4(q2, g.s2); st4(q2 + zplane, g.s4); st4(q2 + 2 * zplane, g.s6); st4(q2 + 3 * zplane, g.s7);
        }
    }
}

// ================================================================================================
// ADJOINT cluster kernel.  One launch runs adjoint steps n_first, n_first-1, ..., n_last.
//   registers : the five adjoint fields of a thread's cells, the five gradient accumulators, the
//               snapshot terms S1..S5 of the step (requested one phase group ahead)
//   LDS       : four planes holding E1..E4 (S^T half) and then D1..D4 (V^T half) on the slab + 2 halo
//               rows, the adjoint memory variables of the slab's C-PML cells, the C-PML tables
//   hand-off  : E2,E3 after the pointwise phase A and D2,D4 after the pointwise phase C (the planes a
//               z-stencil reads); both are functions of the owner's registers only, so they are
//               published before any stencil runs and travel while the interior rows are updated.
// Per step:  A (E, publish, grad_f) | B interior, receive E, B boundary | [receiver injection through
// planes 0,1] | C (D, publish, gradients) | D interior, receive D, D boundary.
// Arithmetic per cell = el_adj_s / el_adj_v.
// ================================================================================================
// Adjoint sources of the single-launch adjoint ("direct" mode): the slab's receiver taps write w g straight into a
// small LDS buffer of their own ([2][kEaRcvRows][PL]: the rows of the slab that hold receivers, compacted) ahead of
// barrier 1, and the owners of those rows add it to v_bar after barrier 3 - no zeroing, no barrier, no atomic.
// It replaced "zero two planes, barrier, ds_add_f32, barrier, read": an LDS float atomic took the receiver slab 1700
// clocks to drain and the two extra barriers 1100 more, per step, in the slab every other slab waits for.  Needs every
// tap of the slab in a cell of its own and at most kEaRcvRows receiver rows (checked once, in LDS, at kernel start);
// anything else keeps the atomic path.
constexpr int kEaRcvRows = 4;
// dynamic LDS the adjoint time loop may ask for: the CU's 160 KB less the kernel's static 256 B and a margin (100x300 in
// 8 slabs takes 144 KB of planes, tables and memory variables + 10 KB of receiver rows)
constexpr int kEaLdsLimit = 158 * 1024;

struct EaParams {
    int nz, nx, ng, gp, pitch;
    unsigned field_stride;
    long long shot_stride;
    int nshot, NW, PL, shot0, shot1;
    int n_first, n_last;                 // steps n_first, n_first-1, ..., n_last
    int nt;
    int W, wl, xr0, wx, fsurf;
    int zrows_max;                       // LDS rows reserved for the z-strip memory variables
    long long psix_shot, psiz_shot, psix_elems;
    const float *mat, *pz, *px;
    float *fields;                       // adjoint fields, global state
    float *psiA, *psiB;                  // ping-pong buffers of the per-step path (absolute in n)
    const float *S;                      // snapshots of step n at S + (n - s_first) * s_step
    int s_first;
    long long s_step;
    float *acc;                          // [nshot][5][nz][gp]
    int nsrc, nrec;
    const int *src_cell;
    const float *src_w;
    float *grad_f;                       // [nt][nshot][nsrc] or null
    const int *rec_cell;
    const float *rec_w, *g_vx, *g_vz;    // g [nt][nshot][nrec]
    const int *slab_cnt, *slab_list;     // receivers per slab: [nshot][NW], [nshot][NW][nrec]
    int rcv_direct;                      // adjoint sources through the receiver-row buffer where the taps allow (kEaRcvRows)
    unsigned long long *xbuf;
    int *err;
    int *xcc_tab;                        // [nshot][NW] XCC_ID + 1 of each slab's workgroup (mifwi::same_xcd)
    int dbg, nap;
    FdK K;                               // stencil weights (fd_order)
#ifdef MIFWI_ABLATIONS
    long long *trace;                    // phase time stamps of one workgroup (MIFWI_EL_CL_TRACE), see EC_STAMP
#endif
};

// receivers of each slab (adjoint sources), one block per shot
__global__ void ec_build_slab_lists(const int *rec_cell, int nrec, int nz, int nx, int NW, int *slab_cnt,
                                    int *slab_list)
{
    const int s = blockIdx.x;
    __shared__ int cnt[64];
    if ((int)threadIdx.x < NW) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int base = nz / NW, rem = nz - base * NW;
    for (int e = threadIdx.x; e < nrec; e += blockDim.x) {
        const int cell = rec_cell[(long long)s * nrec + e];
        if (cell < 0) continue;
        const int i0 = cell / nx;
        const int w = (i0 < rem * (base + 1)) ? i0 / (base + 1) : rem + (i0 - rem * (base + 1)) / base;
        const int pos = atomicAdd(&cnt[w], 1);
        slab_list[((long long)s * NW + w) * nrec + pos] = e;
    }
    __syncthreads();
    if ((int)threadIdx.x < NW) slab_cnt[s * NW + threadIdx.x] = cnt[threadIdx.x];
}

// Where the snapshot planes of the coming phase C are requested (80 KB per workgroup and step, from HBM): slot 0 right
// after the poll of the previous step's phase D (0), the other slots at the head of the B boundary update (2) - half
// the burst at either point.  All of it after the poll, as in round 1, kept the wave's memory queue busy for 3800 clocks
// and delayed the material loads of phase A behind it (vector memory returns in order); measured on 100x300,
// adjoint us per step: (0,0) 10.6, (0,1) 10.2, (1,1) 10.5, (1,2) 10.1, (2,2) 10.3, (0,2) 9.95.
#ifndef EA_S0
#define EA_S0 0
#endif
#ifndef EA_S1
#define EA_S1 2
#endif

struct EaGroup {
    int cls;                                      // 0: none, 1: interior rows of the slab, 2: boundary rows
    int nb;                                       // ec_nb: how the x-neighbours' cells reach this lane
    int g, j, lo;                                 // group in the row, grid row, LDS float offset inside a plane
    int xsl, zsl;                                 // LDS float offset of the group's psi-bar slot, or -1
    float4 bxx, bzz, bxz, vx, vz;                 // adjoint fields
    float4 a0, a1, a2, a3, a4;                    // gradient accumulators (M_L, M_M, M_MU, M_BX, M_BZ order = index)
    float4 S1, S2, S3, S4, S5;                    // snapshot terms of the current step
    int src;                                      // (source index << 2) | cell, or -1
    float src_wt;
};

template <int NG, bool AG, bool XH = false>
__global__ __launch_bounds__(kEcThreads) void el_cluster_adj(const EaParams p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FdK K = p.K;
    const int L = (int)blockIdx.x;
    const int xcd = L & 7, kq = L >> 3;
    const int w = kq % p.NW, s = p.shot0 + xcd + 8 * (kq / p.NW);
    if (s >= p.shot1) return;
    // ablation builds only: slab 1 of the first shot never shows up (a workgroup that was not resident in time)
    if ((kDbg(p) & 64) && w == 1 && s == p.shot0) return;
    const int t = (int)threadIdx.x;
    if (!AG && !mifwi::same_xcd(p.xcc_tab, s, p.NW, w, t, p.err, kEcMaxSpin, kDbg(p) & 128)) return;
    int r0, R;
    ec_slab_rows(p.nz, p.NW, w, r0, R);
    const int PL = p.PL, LR = R + 4;
    const int fsz = LR * PL;
    // 4 planes [LR][PL]: E1 E2 E3 E4 in the S^T half, D1 D2 D4 D3 in the V^T half - in both the plane a
    // backward difference reads across slabs (E3, D4) sits one plane after the forward one (E2, D2)
    float *pln = lds;
    float *lpx = lds + 4 * fsz, *lpz = lpx + 6 * p.gp;        // C-PML tables (ec_stage_tables)
    float *lxs = lpz + 8 * R;                                 // psi-bar of the x strips [4][R][wx]
    float *lzs = lxs + 4 * R * p.wx;                          // psi-bar of the z strips [4][zrows][gp]
    const unsigned ncell = (unsigned)p.nz * p.gp;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long xplane = (long long)p.nz * p.wx, zplane = 2LL * p.W * p.gp;
    // z-strip rows of this slab: top rows [r0, ztop_end), bottom rows [zbot_beg, r0+R)
    const int ztop_end = p.W > 0 ? min(r0 + R, p.W) : r0;
    const int ntop = max(0, ztop_end - r0);
    const int zbot_beg = p.W > 0 ? max(r0, p.nz - p.W) : r0 + R;
    const int zrows = ntop + max(0, r0 + R - zbot_beg);
    const int xsz = R * p.wx, zsz = zrows * p.gp;             // floats per memory variable
    const int PL_ = PL, fsz_ = fsz, xsz_ = xsz, zsz_ = zsz;   // the phase bodies take these through ec_su()
    const unsigned ncell_ = ncell;

    // the buffer the per-step path would READ at step n_first, and the one it would read next
    const int par_in = (p.nt - 1 - p.n_first) & 1;
    const float *psi_in = par_in ? p.psiB : p.psiA;
    const int par_out = (p.nt - 1 - (p.n_last - 1)) & 1;
    float *psi_out = par_out ? p.psiB : p.psiA;

    EaGroup G[NG];
    bool slow = false;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        EaGroup &g = G[q];
        const EcSlot sl = ec_slot(t + q * kEcThreads, R, p.ng, NG * kEcThreads);
        const bool own = sl.cls != 0;
        const int lrw = sl.lrw;
        g.g = sl.g;
        g.j = r0 + lrw;
        g.lo = (lrw + 2) * PL + 4 + 4 * g.g;
        g.cls = sl.cls;
        g.nb = XH ? ec_nb(t + q * kEcThreads, R, p.ng, NG * kEcThreads) : 0;
        g.bxx = g.bzz = g.bxz = g.vx = g.vz = zero4;
        g.a0 = g.a1 = g.a2 = g.a3 = g.a4 = zero4;
        g.S1 = g.S2 = g.S3 = g.S4 = g.S5 = zero4;
        g.xsl = -1; g.zsl = -1; g.src = -1; g.src_wt = 0.f;
        if (own) {
            const unsigned gcc = (unsigned)g.j * p.gp + 4 * g.g;
            const long long o = (long long)(g.j + 2) * p.pitch + 4 + 4 * g.g;
            const float *gf = p.fields + (long long)s * p.shot_stride;
            g.vx = ld4(gf + F_VX * p.field_stride + o); g.vz = ld4(gf + F_VZ * p.field_stride + o);
            g.bxx = ld4(gf + F_SXX * p.field_stride + o); g.bzz = ld4(gf + F_SZZ * p.field_stride + o);
            g.bxz = ld4(gf + F_SXZ * p.field_stride + o);
            const float *ga = p.acc + (long long)s * 5 * ncell + gcc;
            g.a0 = ld4(ga); g.a1 = ld4(ga + ncell); g.a2 = ld4(ga + 2 * (long long)ncell);
            g.a3 = ld4(ga + 3 * (long long)ncell); g.a4 = ld4(ga + 4 * (long long)ncell);
            if (p.W > 0) {
                const int c0 = 4 * g.g;
                int xs_off = -1, zs = -1;
                if (c0 < p.wl) xs_off = c0; else if (c0 >= p.xr0) xs_off = p.wl + (c0 - p.xr0);
                if (g.j < p.W) zs = g.j; else if (g.j >= p.nz - p.W) zs = g.j - (p.nz - 2 * p.W);
                if (xs_off >= 0) {
                    g.xsl = lrw * p.wx + xs_off;
                    const float *qx = psi_in + (long long)s * p.psix_shot + (long long)g.j * p.wx + xs_off;
#pragma unroll
                    for (int k = 0; k < 4; ++k) st4(lxs + k * xsz + g.xsl, ld4(qx + k * xplane));
                }
                if (zs >= 0) {
                    const int zl = (g.j < p.W) ? g.j - r0 : ntop + (g.j - zbot_beg);
                    g.zsl = zl * p.gp + 4 * g.g;
                    const float *qz = psi_in + p.psix_elems + (long long)s * p.psiz_shot + (long long)zs * p.gp + 4 * g.g;
#pragma unroll
                    for (int k = 0; k < 4; ++k) st4(lzs + k * zsz + g.zsl, ld4(qz + k * zplane));
                }
            }
        }
        if (p.grad_f != nullptr)
            for (int e = 0; e < p.nsrc; ++e) {
                const int cell = p.src_cell[(long long)s * p.nsrc + e];
                if (cell < 0) continue;
                const int i0 = cell / p.nx, i1 = cell - i0 * p.nx;
                if (own && i0 == g.j && (i1 >> 2) == g.g) {
                    if (g.src >= 0) slow = true;
                    g.src = (e << 2) | (i1 & 3); g.src_wt = p.src_w[(long long)s * p.nsrc + e];
                }
            }
    }
    slow = __syncthreads_or(slow ? 1 : 0) != 0;
    const bool zero_tap = p.grad_f != nullptr && w == 0 && t < p.nsrc && p.src_cell[(long long)s * p.nsrc + t] < 0;
    // my receiver (adjoint source) of this slab: LDS offset inside a plane + weight; amplitudes are
    // fetched one step ahead
    const int cnt = p.slab_cnt[s * p.NW + w];
    const bool inj_fast = cnt <= kEcThreads;
    int inj_lo = -1, inj_row = 0, inj_col = 0;
    unsigned inj_id4 = 0;
    float inj_w = 0.f, amp_x = 0.f, amp_z = 0.f;
    if (inj_fast && t < cnt) {
        const int inj_id = p.slab_list[((long long)s * p.NW + w) * p.nrec + t];
        const int cell = p.rec_cell[(long long)s * p.nrec + inj_id];
        const int i0 = cell / p.nx, i1 = cell - i0 * p.nx;
        inj_row = i0 - r0; inj_col = i1;
        inj_lo = (i0 - r0 + 2) * PL + 4 + i1;
        inj_w = p.rec_w[(long long)s * p.nrec + inj_id];
        inj_id4 = 4u * inj_id;
    }

    // ---- tables; planes zeroed once (halo rows/columns outside the grid stay zero) -----------------
    ec_stage_tables(lpx, lpz, p.px, p.pz, p.gp, p.nz, r0, R, t);
    for (int e = t; e < fsz; e += kEcThreads) { pln[e] = 0.f; pln[fsz + e] = 0.f; pln[2 * fsz + e] = 0.f; pln[3 * fsz + e] = 0.f; }
    // ---- direct mode of the adjoint sources (kEaRcvRows): row map, tap uniqueness ------------------------------
    int *rowmap = reinterpret_cast<int *>(lzs + 4 * zsz);     // [R (+ pad)]: compact receiver row of a slab row, or -1
    float *rbuf = lzs + 4 * zsz + ((R + 4) & ~3);             // [2][kEaRcvRows][PL]
    int inj_co = -1;                                          // my tap's offset in rbuf (direct mode)
    bool direct = false;
    if (cnt > 0 && inj_fast && p.rcv_direct) {                // MIFWI_EL_ADJ_DIRECT=0 keeps the atomic path
        for (int e = t; e <= R; e += kEcThreads) rowmap[e] = 0;
        for (int e = t; e < 2 * kEaRcvRows * PL; e += kEcThreads) rbuf[e] = 0.f;
        __syncthreads();
        if (t < cnt) rowmap[inj_row] = 1;
        __syncthreads();
        if (t == 0) {
            int k = 0;
            for (int r = 0; r < R; ++r) rowmap[r] = rowmap[r] ? k++ : -1;
            rowmap[R] = k;
        }
        __syncthreads();
        direct = rowmap[R] <= kEaRcvRows;
        if (direct) {
            int *ib = reinterpret_cast<int *>(rbuf);
            if (t < cnt) {
                inj_co = rowmap[inj_row] * PL + 4 + inj_col;
                ib[inj_co] = t;
            }
            __syncthreads();
            const int dup = (t < cnt && ib[inj_co] != t) ? 1 : 0;
            direct = __syncthreads_or(dup) == 0;
            if (t < cnt) ib[inj_co] = 0;                      // = 0.0f
            if (!direct) inj_co = -1;
        }
    }
    __syncthreads();

    // ---- halo hand-off: kind 0 = E2 (plane 1), E3 (plane 2); kind 1 = D2 (plane 1), D4 (plane 2) ----
    EcHandoff X;
    X.init(p.xbuf, s, p.NW, w, R, PL, p.gp, p.ng, t, p.err, false, p.nap, fsz);
    const bool do_x = p.NW > 1 && !(kDbg(p) & 1);
    EcHandoff::Pending P;
    auto complete = [&](unsigned epoch) { X.complete(P, epoch, [&](int off, float v) { (pln + fsz)[off] = v; }); };
    // byte offset of a group inside a [nz][gp] plane (snapshots, materials)
    auto cell_bytes = [&](int jq, int gq) { return 4u * (unsigned)(jq * p.gp + 4 * gq); };
    const float *S_shot = p.S + (long long)s * 5 * ncell;           // step n: S_shot + (n - s_first) * s_step
    auto request_S = [&](EaGroup &g, int n) {
        if (ec_opaque(g.cls) == 0 || (kDbg(p) & 2)) return;
        const unsigned ncell = ec_su(ncell_);
        const float *Sn = S_shot + (long long)(n - p.s_first) * p.s_step;
        const unsigned gcb = cell_bytes(ec_opaque(g.j), ec_opaque(g.g));
        g.S1 = mifwi::ldnt4(ec_at(Sn, gcb)); g.S2 = mifwi::ldnt4(ec_at(Sn + (long long)ncell, gcb));
        g.S3 = mifwi::ldnt4(ec_at(Sn + 2 * (long long)ncell, gcb));
        g.S4 = mifwi::ldnt4(ec_at(Sn + 3 * (long long)ncell, gcb)); g.S5 = mifwi::ldnt4(ec_at(Sn + 4 * (long long)ncell, gcb));
    };
    auto request_amp = [&](int n) {
        if (ec_opaque(inj_lo) >= 0 && n >= p.n_last) {
            const long long o = ((long long)n * p.nshot + s) * p.nrec;
            amp_x = *ec_at(p.g_vx + o, ec_opaque(inj_id4));
            amp_z = *ec_at(p.g_vz + o, ec_opaque(inj_id4));
        }
    };

    // ---- phase bodies ----------------------------------------------------------------------------------
    // A: E from the owner's sigma-bar through the transposed C-PML -> planes; boundary rows publish E2,E3
    // The material planes of BOTH group slots are requested before either is used (EA_MATS_AHEAD): they come from L2,
    // 600-800 clocks away, and a phase that loaded them group by group paid that twice per wave.
    auto mats_a = [&](EaGroup &g, float4 &Ls, float4 &Ms, float4 &mus) {
        const unsigned ncell = ec_su(ncell_);
        const unsigned gcb = cell_bytes(ec_opaque(g.j), ec_opaque(g.g));
        Ls = ld4(ec_at(p.mat + M_L * ncell, gcb)); Ms = ld4(ec_at(p.mat + M_M * ncell, gcb));
        mus = ld4(ec_at(p.mat + M_MU * ncell, gcb));
    };
    auto mats_c = [&](EaGroup &g, float4 &bxs, float4 &bzs) {
        const unsigned ncell = ec_su(ncell_);
        const unsigned gcb = cell_bytes(ec_opaque(g.j), ec_opaque(g.g));
        bxs = ld4(ec_at(p.mat + M_BX * ncell, gcb)); bzs = ld4(ec_at(p.mat + M_BZ * ncell, gcb));
    };
    auto phase_a = [&](EaGroup &g, int n, int it, const int cls, const float4 &Ls, const float4 &Ms, const float4 &mus) {
        const int fsz = ec_su(fsz_), xsz = ec_su(xsz_), zsz = ec_su(zsz_);
        const int lo = ec_opaque(g.lo), gq = ec_opaque(g.g), jq = ec_opaque(g.j);
        // adjoint of szz(0,.) is discarded.  Component-wise selects: a whole-vector select was compiled
        // into a two-entry table in scratch memory (a vector-memory load per use)
        const bool top = p.fsurf && jq == 0;
        const float4 bzz = make_float4(top ? 0.f : g.bzz.x, top ? 0.f : g.bzz.y, top ? 0.f : g.bzz.z,
                                       top ? 0.f : g.bzz.w);
        const int src = ec_opaque(g.src);
        if (p.grad_f != nullptr && src >= 0) {
            float *out = p.grad_f + ((long long)n * p.nshot + s) * p.nsrc;
            // sxx + szz of the four cells, selected with compile-time lane indices: a run-time index into a
            // float4 would move the whole field into scratch memory (vector-memory traffic in the time loop)
            const float pr[4] = {g.bxx.x + bzz.x, g.bxx.y + bzz.y, g.bxx.z + bzz.z, g.bxx.w + bzz.w};
            if (!slow) {
                float v = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c == (src & 3)) v = pr[c];
                *ec_at(out, 4u * (unsigned)(src >> 2)) = fmaf(g.src_wt, v, 0.f);
            } else {                                      // several sources in this group: rescan
                for (int e = 0; e < p.nsrc; ++e) {
                    const int cell = p.src_cell[(long long)s * p.nsrc + e];
                    if (cell < 0) continue;
                    const int i0 = cell / p.nx, i1 = cell - i0 * p.nx;
                    if (i0 != jq || (i1 >> 2) != gq) continue;
                    float v = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c == (i1 & 3)) v = pr[c];
                    out[e] = fmaf(p.src_w[(long long)s * p.nsrc + e], v, 0.f);
                }
            }
        }
        float e1[4], e2[4], e3[4], e4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            e1[c] = fmaf(comp(Ms, c), comp(g.bxx, c), comp(Ls, c) * comp(bzz, c));
            e2[c] = fmaf(comp(Ls, c), comp(g.bxx, c), comp(Ms, c) * comp(bzz, c));
            e3[c] = comp(mus, c) * comp(g.bxz, c);
            e4[c] = e3[c];
        }
        const int xsl = ec_opaque(g.xsl), zsl = ec_opaque(g.zsl);
        if (xsl >= 0) {
            const EcTabX T = ec_tab_x(lpx, 24 * gq);
            float *sl = lxs + xsl;
            const float4 s5 = ld4(sl + 2 * xsz), s8 = ld4(sl + 3 * xsz);
            float n5[4], n8[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                e1[c] = pmlT(comp(s5, c), comp(T.v[PA], c), comp(T.v[PB], c), comp(T.v[PK], c), e1[c], n5[c]);
                e4[c] = pmlT(comp(s8, c), comp(T.v[PAH], c), comp(T.v[PBH], c), comp(T.v[PKH], c), e4[c], n8[c]);
            }
            st4(sl + 2 * xsz, make_float4(n5[0], n5[1], n5[2], n5[3]));
            st4(sl + 3 * xsz, make_float4(n8[0], n8[1], n8[2], n8[3]));
        }
        if (zsl >= 0) {
            const EcTabZ Z = ec_tab_z(lpz, 8 * (jq - r0));
            float *sl = lzs + zsl;
            const float4 s6 = ld4(sl + 2 * zsz), s7 = ld4(sl + 3 * zsz);
            float n6[4], n7[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                e2[c] = pmlT(comp(s6, c), Z.v[PA], Z.v[PB], Z.v[PK], e2[c], n6[c]);
                e3[c] = pmlT(comp(s7, c), Z.v[PAH], Z.v[PBH], Z.v[PKH], e3[c], n7[c]);
            }
            st4(sl + 2 * zsz, make_float4(n6[0], n6[1], n6[2], n6[3]));
            st4(sl + 3 * zsz, make_float4(n7[0], n7[1], n7[2], n7[3]));
        }
        const float4 E2 = make_float4(e2[0], e2[1], e2[2], e2[3]), E3 = make_float4(e3[0], e3[1], e3[2], e3[3]);
        st4(pln + lo, make_float4(e1[0], e1[1], e1[2], e1[3]));
        st4(pln + fsz + lo, E2);
        st4(pln + 2 * fsz + lo, E3);
        st4(pln + 3 * fsz + lo, make_float4(e4[0], e4[1], e4[2], e4[3]));
        if (do_x && cls == 2) X.template publish<AG>(jq - r0, gq, 0, (unsigned)(2 * it + 1), it & 1, E2, E3);
    };
    // B: v_bar -= stencils(E)
    auto phase_b = [&](EaGroup &g) {
        const int fsz = ec_su(fsz_), PL = ec_su(PL_);
        const int lo = ec_opaque(g.lo), gq = ec_opaque(g.g);
        const float *E1 = pln + lo, *E2 = pln + fsz + lo, *E3 = pln + 2 * fsz + lo, *E4 = pln + 3 * fsz + lo;
        const int nb = XH ? ec_opaque(g.nb) : 0;
        const float4 c1 = ld4(E1);
        const float4 c4 = ld4(E4);
        const float4 t0 = ld4(E3 - 2 * PL), t1 = ld4(E3 - PL), t2 = ld4(E3), t3 = ld4(E3 + PL);
        const float4 u0 = ld4(E2 - PL), u1 = ld4(E2), u2 = ld4(E2 + PL), u3 = ld4(E2 + 2 * PL);
        float2 L1, R1, L4, R4;
        ec_xhalo<XH>(E1, c1, nb, pln, L1, R1);
        ec_xhalo<XH>(E4, c4, nb, pln, L4, R4);
        const float x1[8] = {L1.x, L1.y, c1.x, c1.y, c1.z, c1.w, R1.x, R1.y};
        const float x4[8] = {L4.x, L4.y, c4.x, c4.y, c4.z, c4.w, R4.x, R4.y};
        float nvx[4], nvz[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float dx1 = dfw(K, x1[c + 1], x1[c + 2], x1[c + 3], x1[c + 4]);
            const float dz3 = dbw(K, comp(t0, c), comp(t1, c), comp(t2, c), comp(t3, c));
            const float dz2 = dfw(K, comp(u0, c), comp(u1, c), comp(u2, c), comp(u3, c));
            const float dx4 = dbw(K, x4[c], x4[c + 1], x4[c + 2], x4[c + 3]);
            nvx[c] = comp(g.vx, c) - (dx1 + dz3);
            nvz[c] = comp(g.vz, c) - (dz2 + dx4);
            if (4 * gq + c >= p.nx) { nvx[c] = 0.f; nvz[c] = 0.f; }
        }
        g.vx = make_float4(nvx[0], nvx[1], nvx[2], nvx[3]);
        g.vz = make_float4(nvz[0], nvz[1], nvz[2], nvz[3]);
    };
    // C: D from the new v_bar -> planes [D1 D2 D4 D3]; boundary rows publish D2,D4; all five gradient accumulators
    auto phase_c = [&](EaGroup &g, int it, const int cls, const float4 &bxs, const float4 &bzs) {
        const int fsz = ec_su(fsz_), xsz = ec_su(xsz_), zsz = ec_su(zsz_);
        const int lo = ec_opaque(g.lo), gq = ec_opaque(g.g), jq = ec_opaque(g.j);
        float d1[4], d2[4], d3[4], d4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            d1[c] = comp(bxs, c) * comp(g.vx, c); d2[c] = d1[c];
            d3[c] = comp(bzs, c) * comp(g.vz, c); d4[c] = d3[c];
        }
        const int xsl = ec_opaque(g.xsl), zsl = ec_opaque(g.zsl);
        if (xsl >= 0) {
            const EcTabX T = ec_tab_x(lpx, 24 * gq);
            float *sl = lxs + xsl;
            const float4 s1 = ld4(sl), s3 = ld4(sl + xsz);
            float n1[4], n3[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                d1[c] = pmlT(comp(s1, c), comp(T.v[PAH], c), comp(T.v[PBH], c), comp(T.v[PKH], c), d1[c], n1[c]);
                d3[c] = pmlT(comp(s3, c), comp(T.v[PA], c), comp(T.v[PB], c), comp(T.v[PK], c), d3[c], n3[c]);
            }
            st4(sl, make_float4(n1[0], n1[1], n1[2], n1[3]));
            st4(sl + xsz, make_float4(n3[0], n3[1], n3[2], n3[3]));
        }
        if (zsl >= 0) {
            const EcTabZ Z = ec_tab_z(lpz, 8 * (jq - r0));
            float *sl = lzs + zsl;
            const float4 s2 = ld4(sl), s4 = ld4(sl + zsz);
            float n2[4], n4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                d2[c] = pmlT(comp(s2, c), Z.v[PA], Z.v[PB], Z.v[PK], d2[c], n2[c]);
                d4[c] = pmlT(comp(s4, c), Z.v[PAH], Z.v[PBH], Z.v[PKH], d4[c], n4[c]);
            }
            st4(sl, make_float4(n2[0], n2[1], n2[2], n2[3]));
            st4(sl + zsz, make_float4(n4[0], n4[1], n4[2], n4[3]));
        }
        const float4 D2 = make_float4(d2[0], d2[1], d2[2], d2[3]), D4 = make_float4(d4[0], d4[1], d4[2], d4[3]);
        st4(pln + lo, make_float4(d1[0], d1[1], d1[2], d1[3]));
        st4(pln + fsz + lo, D2);
        st4(pln + 2 * fsz + lo, D4);
        st4(pln + 3 * fsz + lo, make_float4(d3[0], d3[1], d3[2], d3[3]));
        if (do_x && cls == 2) X.template publish<AG>(jq - r0, gq, 1, (unsigned)(2 * it + 2), it & 1, D2, D4);
        // gradients (oracle order): Ms, Ls, mus from the old sigma_bar; bxs, bzs from the new v_bar
        const bool top = p.fsurf && jq == 0;
        const float4 bzz = make_float4(top ? 0.f : g.bzz.x, top ? 0.f : g.bzz.y, top ? 0.f : g.bzz.z,
                                       top ? 0.f : g.bzz.w);
#define EA_ACC3(dst, a, b, c_, d) dst = fmaf(a, b, fmaf(c_, d, dst))
        EA_ACC3(g.a1.x, g.S1.x, g.bxx.x, g.S2.x, bzz.x); EA_ACC3(g.a1.y, g.S1.y, g.bxx.y, g.S2.y, bzz.y);
        EA_ACC3(g.a1.z, g.S1.z, g.bxx.z, g.S2.z, bzz.z); EA_ACC3(g.a1.w, g.S1.w, g.bxx.w, g.S2.w, bzz.w);
        EA_ACC3(g.a0.x, g.S2.x, g.bxx.x, g.S1.x, bzz.x); EA_ACC3(g.a0.y, g.S2.y, g.bxx.y, g.S1.y, bzz.y);
        EA_ACC3(g.a0.z, g.S2.z, g.bxx.z, g.S1.z, bzz.z); EA_ACC3(g.a0.w, g.S2.w, g.bxx.w, g.S1.w, bzz.w);
#undef EA_ACC3
        g.a2.x = fmaf(g.S3.x, g.bxz.x, g.a2.x); g.a2.y = fmaf(g.S3.y, g.bxz.y, g.a2.y);
        g.a2.z = fmaf(g.S3.z, g.bxz.z, g.a2.z); g.a2.w = fmaf(g.S3.w, g.bxz.w, g.a2.w);
        g.a3.x = fmaf(g.S4.x, g.vx.x, g.a3.x); g.a3.y = fmaf(g.S4.y, g.vx.y, g.a3.y);
        g.a3.z = fmaf(g.S4.z, g.vx.z, g.a3.z); g.a3.w = fmaf(g.S4.w, g.vx.w, g.a3.w);
        g.a4.x = fmaf(g.S5.x, g.vz.x, g.a4.x); g.a4.y = fmaf(g.S5.y, g.vz.y, g.a4.y);
        g.a4.z = fmaf(g.S5.z, g.vz.z, g.a4.z); g.a4.w = fmaf(g.S5.w, g.vz.w, g.a4.w);
    };
    // D: sigma_bar -= stencils(D) (+ transposed free-surface mirroring on grid rows 0 and 1)
    auto phase_d = [&](EaGroup &g, const bool edge) {
        const int fsz = ec_su(fsz_), PL = ec_su(PL_);
        const int lo = ec_opaque(g.lo), gq = ec_opaque(g.g);
        const float *D1 = pln + lo, *D2 = pln + fsz + lo, *D4 = pln + 2 * fsz + lo, *D3 = pln + 3 * fsz + lo;
        const int nb = XH ? ec_opaque(g.nb) : 0;
        const float4 c1 = ld4(D1);
        const float4 c3 = ld4(D3);
        const float4 u0 = ld4(D2 - PL), u1 = ld4(D2), u2 = ld4(D2 + PL), u3 = ld4(D2 + 2 * PL);
        const float4 t0 = ld4(D4 - 2 * PL), t1 = ld4(D4 - PL), t2 = ld4(D4), t3 = ld4(D4 + PL);
        float2 L1, R1, L3, R3;
        ec_xhalo<XH>(D1, c1, nb, pln, L1, R1);
        ec_xhalo<XH>(D3, c3, nb, pln, L3, R3);
        const float x1[8] = {L1.x, L1.y, c1.x, c1.y, c1.z, c1.w, R1.x, R1.y};
        const float x3[8] = {L3.x, L3.y, c3.x, c3.y, c3.z, c3.w, R3.x, R3.y};
        float nxx[4], nzz[4], nxz[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float dx1 = dbw(K, x1[c], x1[c + 1], x1[c + 2], x1[c + 3]);
            const float dz2 = dfw(K, comp(u0, c), comp(u1, c), comp(u2, c), comp(u3, c));
            const float dx3 = dfw(K, x3[c + 1], x3[c + 2], x3[c + 3], x3[c + 4]);
            const float dz4 = dbw(K, comp(t0, c), comp(t1, c), comp(t2, c), comp(t3, c));
            nxx[c] = comp(g.bxx, c) - dx1;
            nxz[c] = comp(g.bxz, c) - (dz2 + dx3);
            nzz[c] = comp(g.bzz, c) - dz4;
        }
        if (edge && p.fsurf) {
            const int jq = ec_opaque(g.j);
            if (jq < 2) {
                // grid rows 0 and 1 are local rows 2 and 3 of slab 0
                const float4 r0d2 = ld4(pln + fsz + 2 * PL + 4 + 4 * gq), r1d2 = ld4(pln + fsz + 3 * PL + 4 + 4 * gq);
                const float4 r0d4 = ld4(pln + 2 * fsz + 2 * PL + 4 + 4 * gq);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (jq == 0) nxz[c] = nxz[c] + fmaf(K.c1, comp(r0d2, c), K.c2 * comp(r1d2, c));
                    else { nxz[c] = nxz[c] + K.c2 * comp(r0d2, c); nzz[c] = nzz[c] + K.c2 * comp(r0d4, c); }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (4 * gq + c >= p.nx) { nxx[c] = 0.f; nxz[c] = 0.f; nzz[c] = 0.f; }
        g.bxx = make_float4(nxx[0], nxx[1], nxx[2], nxx[3]);
        g.bzz = make_float4(nzz[0], nzz[1], nzz[2], nzz[3]);
        g.bxz = make_float4(nxz[0], nxz[1], nxz[2], nxz[3]);
    };

    const int nsteps = p.n_first - p.n_last + 1;
#ifdef MIFWI_ABLATIONS
    const bool tr_on = w == (((kDbg(p) >> 8) & 15) ? ((kDbg(p) >> 8) & 15) - 1 : p.NW / 2) && s == p.shot0;   // dbg bits 8-11: traced slab + 1
#endif
    ec_drain_vmem();
#pragma unroll
    for (int q = 0; q < NG; ++q) request_S(G[q], p.n_first);
    request_amp(p.n_first);
    for (int it = 0; it < nsteps; ++it) {
        const int n = p.n_first - it;
        EC_LAGGARD();
        EC_STAMP(0);
        // ---- A ----------------------------------------------------------------------------------------
        float4 mL[NG], mM[NG], mMu[NG];
#if EA_MATS_AHEAD_A
#pragma unroll
        for (int qq = 0; qq < NG; ++qq) {
            const int q = EA_PUBLISH_FIRST ? NG - 1 - qq : qq;
            if (ec_opaque(G[q].cls) != 0) mats_a(G[q], mL[q], mM[q], mMu[q]);
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int qq = 0; qq < NG; ++qq) {
            const int q = EA_PUBLISH_FIRST ? NG - 1 - qq : qq;     // the second slot holds the rows that publish (ec_slot)
            const int cls = ec_opaque(G[q].cls);
#if !EA_MATS_AHEAD_A
            if (cls != 0) mats_a(G[q], mL[q], mM[q], mMu[q]);
#endif
            if (cls != 0) phase_a(G[q], n, it, cls, mL[q], mM[q], mMu[q]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (p.grad_f != nullptr && w == 0) {               // inactive source taps: slab 0 writes their zeros
            if (zero_tap) (p.grad_f + ((long long)n * p.nshot + s) * p.nsrc)[t] = 0.f;
            for (int e = t + kEcThreads; e < p.nsrc; e += kEcThreads)
                if (p.src_cell[(long long)s * p.nsrc + e] < 0) p.grad_f[((long long)n * p.nshot + s) * p.nsrc + e] = 0.f;
        }
        if (direct) {                                      // this step's adjoint sources: read after barrier 3
            const int co = ec_opaque(inj_co);
            if (co >= 0) {
                rbuf[co] = inj_w * amp_x;
                rbuf[kEaRcvRows * PL_ + co] = inj_w * amp_z;
            }
        }
        EC_STAMP(1);
        __syncthreads();                                   // 1: E planes complete on the own rows
        EC_STAMP(2);
        if (it > 0) {
#pragma unroll
            for (int q = 0; q < NG; ++q)
                if ((q == 0 ? EA_S0 : EA_S1) == 1) request_S(G[q], n);
        }
        // ---- B ----------------------------------------------------------------------------------------
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            if ((!EC_LATE_INTERIOR || q == 0) && ec_opaque(G[q].cls) == 1) phase_b(G[q]);
            __builtin_amdgcn_sched_barrier(0);
        }
        EC_STAMP(3);
        if (do_x) {
            X.request(P, 0, it & 1);
            complete((unsigned)(2 * it + 1));
        }
        EC_STAMP(4);
        __syncthreads();                                   // 2: E halo rows are in LDS
        EC_STAMP(5);
#ifndef MIFWI_EA_NO_SETTLE
        // the adjoint-source amplitudes of this step (requested a step ago, arrived before the poll's drain) must not
        // wait behind the snapshot planes requested next: in the slab that holds the receivers - the slowest one, which
        // every other slab waits for - their first use stalled 3500 clocks on loads it does not need
        ec_settle(amp_x);
        ec_settle(amp_z);
#endif
        if (it > 0) {
#pragma unroll
            for (int q = 0; q < NG; ++q)
                if ((q == 0 ? EA_S0 : EA_S1) == 2) request_S(G[q], n);
        }
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            if (ec_opaque(G[q].cls) >= 2) phase_b(G[q]);                 // boundary + late interior (ec_slot)
            __builtin_amdgcn_sched_barrier(0);
        }
        EC_STAMP(6);
        __syncthreads();                                   // 3: every read of E is done
        EC_STAMP(7);
        // ---- receivers of this slab: v_bar += w g, through planes 0 and 1 ---------------------------
        if (direct) {
#pragma unroll
            for (int q = 0; q < NG; ++q)
                if (ec_opaque(G[q].cls) != 0) {
                    const int rr = rowmap[ec_opaque(G[q].j) - r0];
                    if (rr >= 0) {
                        const float *b = rbuf + rr * PL_ + 4 + 4 * ec_opaque(G[q].g);
                        const float4 ix = ld4(b), iz = ld4(b + kEaRcvRows * PL_);
                        G[q].vx = make_float4(G[q].vx.x + ix.x, G[q].vx.y + ix.y, G[q].vx.z + ix.z, G[q].vx.w + ix.w);
                        G[q].vz = make_float4(G[q].vz.x + iz.x, G[q].vz.y + iz.y, G[q].vz.z + iz.z, G[q].vz.w + iz.w);
                    }
                }
        } else if (cnt > 0) {
#pragma unroll
            for (int q = 0; q < NG; ++q)
                if (ec_opaque(G[q].cls) != 0) { st4(pln + ec_opaque(G[q].lo), zero4); st4(pln + fsz + ec_opaque(G[q].lo), zero4); }
            __syncthreads();
            if (inj_fast) {
                const int il = ec_opaque(inj_lo);
                if (il >= 0) {
                    atomicAdd(pln + il, inj_w * amp_x);
                    atomicAdd(pln + fsz + il, inj_w * amp_z);
                }
            } else {
                const int *lst = p.slab_list + ((long long)s * p.NW + w) * p.nrec;
                for (int e = t; e < cnt; e += kEcThreads) {
                    const int id = lst[e];
                    const int cell = p.rec_cell[(long long)s * p.nrec + id];
                    const int i0 = cell / p.nx, i1 = cell - i0 * p.nx;
                    const float ww = p.rec_w[(long long)s * p.nrec + id];
                    const long long o = ((long long)n * p.nshot + s) * p.nrec + id;
                    atomicAdd(pln + (i0 - r0 + 2) * PL + 4 + i1, ww * p.g_vx[o]);
                    atomicAdd(pln + fsz + (i0 - r0 + 2) * PL + 4 + i1, ww * p.g_vz[o]);
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NG; ++q)
                if (ec_opaque(G[q].cls) != 0) {
                    const float4 ix = ld4(pln + ec_opaque(G[q].lo)), iz = ld4(pln + fsz + ec_opaque(G[q].lo));
                    G[q].vx = make_float4(G[q].vx.x + ix.x, G[q].vx.y + ix.y, G[q].vx.z + ix.z, G[q].vx.w + ix.w);
                    G[q].vz = make_float4(G[q].vz.x + iz.x, G[q].vz.y + iz.y, G[q].vz.z + iz.z, G[q].vz.w + iz.w);
                }
        }
        EC_STAMP(8);
        // ---- C ----------------------------------------------------------------------------------------
        float4 mBx[NG], mBz[NG];
#if EA_MATS_AHEAD_C
#pragma unroll
        for (int qq = 0; qq < NG; ++qq) {
            const int q = EA_PUBLISH_FIRST ? NG - 1 - qq : qq;
            if (ec_opaque(G[q].cls) != 0) mats_c(G[q], mBx[q], mBz[q]);
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int qq = 0; qq < NG; ++qq) {
            const int q = EA_PUBLISH_FIRST ? NG - 1 - qq : qq;
            const int cls = ec_opaque(G[q].cls);
#if !EA_MATS_AHEAD_C
            if (cls != 0) mats_c(G[q], mBx[q], mBz[q]);
#endif
            if (cls != 0) phase_c(G[q], it, cls, mBx[q], mBz[q]);
            __builtin_amdgcn_sched_barrier(0);
        }
        EC_STAMP(9);
        __syncthreads();                                   // 4: D planes complete on the own rows
        EC_STAMP(10);
        // ---- D ----------------------------------------------------------------------------------------
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            if ((!EC_LATE_INTERIOR || q == 0) && ec_opaque(G[q].cls) == 1) phase_d(G[q], false);
            __builtin_amdgcn_sched_barrier(0);
        }
        EC_STAMP(11);
        if (do_x) {
            X.request(P, 1, it & 1);
            complete((unsigned)(2 * it + 2));
        }
        EC_STAMP(12);
        if (!do_x) ec_drain_vmem();                        // one slab: no poll has said so (EcHandoff::complete)
        if (it + 1 < nsteps) {
#pragma unroll
            for (int q = 0; q < NG; ++q)
                if ((q == 0 ? EA_S0 : EA_S1) == 0) request_S(G[q], n - 1);      // after the poll: loads retire in order
        }
        // The next step's adjoint-source amplitudes ride with the snapshot planes: loads retire in order, so the first
        // wait after a request pays its full latency.  Requested where they are used (ahead of phase C's material
        // loads), they cost the slab that holds the receivers - the one every other slab ends up waiting for - a
        // memory round trip per step (3300 of 24 000 clocks).
        if (cnt > 0) request_amp(n - 1);
        EC_STAMP(13);
        __syncthreads();                                   // 5: D halo rows are in LDS
        EC_STAMP(14);
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const int cls = ec_opaque(G[q].cls);
            if (cls == 2) phase_d(G[q], true);
            else if (q > 0 && cls == 3) phase_d(G[q], false);
            __builtin_amdgcn_sched_barrier(0);
        }
        EC_STAMP(15);
        if ((it & 31) == 31 || it == nsteps - 1) {
            if (__syncthreads_or(X.failed ? 1 : 0)) {        // 6 (+ collective time-out check)
                if (t == 0) __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        } else {
            __syncthreads();                               // 6: every read of D is done
        }
    }

    // ---- state back to the layout of the per-step path ------------------------------------------------
    float *gf = p.fields + (long long)s * p.shot_stride;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const EaGroup &g = G[q];
        if (g.cls == 0) continue;
        const long long o = (long long)(g.j + 2) * p.pitch + 4 + 4 * g.g;
        st4(gf + F_VX * p.field_stride + o, g.vx); st4(gf + F_VZ * p.field_stride + o, g.vz);
        st4(gf + F_SXX * p.field_stride + o, g.bxx); st4(gf + F_SZZ * p.field_stride + o, g.bzz);
        st4(gf + F_SXZ * p.field_stride + o, g.bxz);
        const unsigned gcc = (unsigned)g.j * p.gp + 4 * g.g;
        float *ga = p.acc + (long long)s * 5 * ncell + gcc;
        st4(ga, g.a0); st4(ga + ncell, g.a1); st4(ga + 2 * (long long)ncell, g.a2);
        st4(ga + 3 * (long long)ncell, g.a3); st4(ga + 4 * (long long)ncell, g.a4);
        if (g.xsl >= 0) {
            const int xs_off = g.xsl - (g.j - r0) * p.wx;
            float *qx = psi_out + (long long)s * p.psix_shot + (long long)g.j * p.wx + xs_off;
#pragma unroll
            for (int k = 0; k < 4; ++k) st4(qx + k * xplane, ld4(lxs + k * xsz + g.xsl));
        }
        if (g.zsl >= 0) {
            const int zs = (g.j < p.W) ? g.j : g.j - (p.nz - 2 * p.W);
            float *qz = psi_out + p.psix_elems + (long long)s * p.psiz_shot + (long long)zs * p.gp + 4 * g.g;
#pragma unroll
            for (int k = 0; k < 4; ++k) st4(qz + k * zplane, ld4(lzs + k * zsz + g.zsl));
        }
    }
}
