// Shared host-side plumbing of libmifwi (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "../../include/mifwi.h"

namespace mifwi {

inline char *err_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

inline int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define MIFWI_HIP_TRY(expr)                                                              \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess)                                                           \
            return ::mifwi::fail(MIFWI_EHIP, "%s failed: %s (%s:%d)", #expr,             \
                                 hipGetErrorString(e__), __FILE__, __LINE__);            \
    } while (0)

// Internal status of a single-launch time loop whose hand-off timed out (some workgroups were not
// resident in time, e.g. another stream held CUs for a long kernel).  The call is then re-run with one
// launch per (half) step: from the zero state when it started there, otherwise from the copy of its
// input state that resumed calls keep (time checkpointing).
constexpr int kClusterTimedOut = 1;
// ... or whose placement check failed (the slabs of a shot were not dealt to one XCD): the launch is repeated once with
// granules published at agent scope (through the fabric: correct on any placement, 10-15 % slower per step) before the
// per-step kernels are considered.
constexpr int kClusterMisplaced = 2;
// The error block of a single-launch kernel (last 64 ints of its hand-off buffer): [0] time-out raised, [1] placement
// check failed, [2] slow polls: waves x polls that needed more than kSlowPollPasses passes.
constexpr int kErrTimeout = 0, kErrPlacement = 1, kErrSlow = 2;
constexpr unsigned kSlowPollPasses = 32;
// A single-launch time loop gave up (a workgroup was not resident in time, or the slabs of a shot were not placed on
// one XCD) and the call is re-run with one launch per step: correct, but several times slower on small grids, so it
// is said once per process on stderr (MIFWI_QUIET=1 silences it) rather than left to be discovered in a profile.
inline std::atomic<long long> g_fallbacks{0};          // mifwi_fallback_count()
inline void note_fallback(const char *what)
{
    g_fallbacks.fetch_add(1, std::memory_order_relaxed);
    static bool said = false;
    if (said || getenv("MIFWI_QUIET")) return;
    said = true;
    fprintf(stderr, "libmifwi: %s: single-launch time loop gave up (hand-off time-out or XCD placement); "
                    "falling back to one launch per step for this call\n", what);
}

// launches repeated with agent-scope publishes after a failed placement check (mifwi_agent_handoff_count()), and
// launches in which some workgroup waited more than kSlowPollPasses poll passes for a neighbour - the signature of a
// GPU shared with another process (mifwi_slow_handoff_count()): results are unaffected, the speed is not
inline std::atomic<long long> g_agent_tier{0}, g_slow_handoffs{0};
inline void note_agent_tier(const char *what)
{
    g_agent_tier.fetch_add(1, std::memory_order_relaxed);
    static bool said = false;
    if (said || getenv("MIFWI_QUIET")) return;
    said = true;
    fprintf(stderr, "libmifwi: %s: the slabs of a shot were not placed on one XCD; repeating the single-launch time loop "
                    "with hand-offs through the fabric\n", what);
}
inline void note_slow_handoff(const char *what, int workgroups)
{
    g_slow_handoffs.fetch_add(1, std::memory_order_relaxed);
    static bool said = false;
    if (said || getenv("MIFWI_QUIET")) return;
    said = true;
    fprintf(stderr, "libmifwi: %s: %d time(s) a wave of a single-launch time loop waited more than %u poll passes for a "
                    "neighbouring slab - is another process using this GPU?  (results are unaffected; "
                    "mifwi_slow_handoff_count() counts such launches)\n", what, workgroups, kSlowPollPasses);
}
// what the host makes of the error block after a launch
inline int cluster_verdict(const int *e, const char *what)
{
    if (e[kErrSlow] > 0) note_slow_handoff(what, e[kErrSlow]);
    if (e[kErrPlacement] != 0) return kClusterMisplaced;
    return e[kErrTimeout] != 0 ? kClusterTimedOut : MIFWI_OK;
}

// test hook, read per call: MIFWI_TEST_FAKE_TIMEOUT=1 treats every single-launch attempt as timed out before it
// runs, =2 after it has run (the state has been advanced and must really be restored)
inline int fake_timeout()
{
    const char *v = getenv("MIFWI_TEST_FAKE_TIMEOUT");
    return (v && (*v == '1' || *v == '2')) ? *v - '0' : 0;
}

inline int check_device(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(MIFWI_ENODEVICE, "no HIP device visible (libmifwi has no CPU fallback)");
    if (device < 0 || device >= n)
        return fail(MIFWI_ENODEVICE, "device %d out of range (0..%d)", device, n - 1);
    return MIFWI_OK;
}

// Dynamic-LDS ceiling requested for every cluster kernel.  The attribute is per function, not per
// plan: plans of different grids coexist (forward of one model, backward of another), so each
// plan asks for the common cap and passes its own (smaller) size at launch.
constexpr int kClusterLdsLimit = 150 * 1024;      // = the eligibility bound of every cluster plan

// Timing ablations of the single-launch kernels (skip hand-off / snapshot stream / waiting; wrong results)
// exist only in builds with -DMIFWI_ABLATIONS; the shipped kernels do not carry the tests.
#ifdef MIFWI_ABLATIONS
#define kDbg(p) ((p).dbg)
#else
#define kDbg(p) 0
#endif

#ifdef __HIPCC__
// nap between two poll passes of a halo hand-off (s_sleep takes an immediate: four lengths, ~64 clocks each
// unit).  Fat slabs (interior rows cover the flight, a miss means the neighbour is a good microsecond
// behind) nap long - every extra pass is memory traffic in the way of the publishes; thin slabs, whose
// step is the hand-off chain itself, nap short.
__device__ __forceinline__ void poll_nap(int units)
{
    if (units >= 127) __builtin_amdgcn_s_sleep(127);
    else if (units >= 96) __builtin_amdgcn_s_sleep(96);
    else if (units >= 64) __builtin_amdgcn_s_sleep(64);
    else if (units >= 48) __builtin_amdgcn_s_sleep(48);
    else if (units >= 16) __builtin_amdgcn_s_sleep(16);
    else if (units >= 4) __builtin_amdgcn_s_sleep(4);
    else __builtin_amdgcn_s_sleep(1);
}

// Halo hand-offs between the row slabs of a shot stay inside ONE XCD's L2: the granules are published with plain
// (workgroup-scope) stores, which stay in that L2, and polled with agent-scope loads, which bypass the reader's L1 and
// are served by the same L2 - an L2 round trip instead of two trips over the fabric (agent-scope stores leave the
// L2 and drop the line: measured 2000-2500 clocks per poll against 900-1200, elastic 100x300 forward 7.2 -> 6.4 us
// per step).  That is only correct while every slab of a shot runs on the same XCD.  The block -> (shot, slab) map puts
// them on blocks b, b + 8, b + 16, ..., which the dispatcher deals to one XCD - observed, not promised by HIP - so it
// is checked, not assumed: before its time loop every workgroup writes its XCC_ID to xcc_tab[shot][slab] (agent
// scope) and reads its two neighbours' entries; a mismatch or a neighbour that never shows up makes the launch bail
// out through the error block (a mismatch also raises its placement word), and the host repeats the launch with
// agent-scope publishes - the AG variants of the kernels, which skip this check - or, after a no-show, re-runs the range
// with one launch per step.
// (A stale granule can never be taken for a fresh one - its tag is an older epoch - so a wrong placement could only
// ever end in that time-out, not in wrong numbers; the check just gets there in microseconds.)
// Returns true when this workgroup may run; collective over the workgroup (__syncthreads_or inside).
__device__ __forceinline__ bool same_xcd(int *xcc_tab, int s, int NW, int w, int t, int *err, unsigned max_spin, int fake)
{
    if (NW == 1) return true;
    bool bad = false, misplaced = false;
    if (t == 0) {
        // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, 4 bits)
        const int me = fake ? w + 1 : (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15) + 1;
        __hip_atomic_store(xcc_tab + s * NW + w, me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int o = w - 1; o <= w + 1; o += 2) {
            if (o < 0 || o >= NW) continue;
            int v = 0;
            for (unsigned spins = 0; spins < max_spin; ++spins) {
                v = __hip_atomic_load(xcc_tab + s * NW + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v != 0) break;
                if ((spins & 255u) == 255u && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (v != me) bad = true;                     // 0: the neighbour never showed up
            if (v != me && v != 0) misplaced = true;
        }
        if (misplaced) __hip_atomic_store(err + kErrPlacement, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bad) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return __syncthreads_or(bad ? 1 : 0) == 0;
}

// write-once / read-once streams (snapshots): non-temporal accesses keep them out of the way of the
// cache-resident planes
typedef float mifwi_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stnt4(float *p, const float4 &v)
{
    __builtin_nontemporal_store(mifwi_v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<mifwi_v4f *>(p));
}
__device__ __forceinline__ float4 ldnt4(const float *p)
{
    const mifwi_v4f v = __builtin_nontemporal_load(reinterpret_cast<const mifwi_v4f *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
#endif

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int64_t round_up64(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

}  // namespace mifwi
