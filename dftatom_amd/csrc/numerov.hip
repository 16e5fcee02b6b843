// numerov.hip -- batched Numerov shooting sweeps for gfx950 (wave64).
//
// Replaces DFT::Numerov<NumerovFunctionNonUniformGrid>::SolveSchrodingerCountNodes /
// SolveSchrodingerSolutionInZero / SolveSchrodingerMatchSolutionCompletely (Numerov.h:272-504).
//
// Mapping.  A sweep is a strictly sequential three-term recurrence over the grid index, so one LANE owns
// one trial (potential, l, E) and the 64 lanes of a wavefront march the grid index i together (inward,
// i = start-2 .. 1).  All lanes of a wave share (potential, l); the per-point inputs
//     veff_i = V_i + l(l+1)/(r_i r_i)/2      and      e2_i = exp(2 i delta)
// are therefore wave-uniform; only E differs per lane.  Two kernels implement the sweeps with the same arithmetic:
// k_sweep (one wave per block of 64 trials, everything fused; for a full machine) and k_sweep_pipe (one workgroup
// per block, the per-point inputs and the loop-carried recurrence on different SIMDs; for a few hundred blocks).
// Arithmetic is the reference's, in its operation order, IEEE fp64 with contraction off:
//     f_i = 2 (veff_i - E) Rp^2 delta^2 e2_i + delta^2/4           (Numerov.h:96-101)
//     w_i = 2 w_{i+1} - w_{i+2} + u_{i+1} f_{i+1};  u_i = w_i / (1 - f_i/12)   (Numerov.h:311-321,510-513)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <type_traits>
#include <vector>

#include "internal.h"
#include "levels.h"
#include "ordered_sum.h"

namespace {

constexpr double kH2p12 = 1. / 12.;   // Numerov.h:287

struct GridScalars {
    int N;
    double delta, Rp2delta2, delta2p4, far_thr;
    int uniform;                  // NumerovFunctionRegularGrid (Numerov.h:16-70)
    double Rmax, h, h2, h2p12;    // uniform: step of the grid and of the recurrence (Numerov.h:276-278)
};

__device__ __forceinline__ double f_of(double veff, double e2, double E, const GridScalars& gs)
{
    // Numerov.h:100: 2. * (effectivePotential - E) * Rp2delta2 * exp(posIndex * twodelta) + delta2p4
    return 2. * (veff - E) * gs.Rp2delta2 * e2 + gs.delta2p4;
}

// (A pointer that reaches device code through a struct or a call, not as a kernel argument, is a generic one to the compiler: flat loads
// and stores.  The out-of-line parts of the device-side search re-derive theirs through dfta_as_global, common.h.)

// ---- table of wave-uniform per-point inputs -----------------------------------------------------------
// tab[(slot)*N + i] = { V[v][i] + cl[l][i], e2[i] }   (Numerov.h:93: V + l(l+1)/(r r) * 0.5)
// uniform grid: .y carries V_i itself (f needs no exp factor there, and the first two points of a sweep and the whole
// match solve evaluate the centrifugal term at positions that are not i h: Numerov.h:289-296, 430-458)
__global__ void k_build_tab(double2* __restrict__ tab, const double* __restrict__ V, const double* __restrict__ cl,
                            const double* __restrict__ e2, const int* __restrict__ slot_v, const int* __restrict__ slot_l, int N,
                            int uniform)
{
    const int slot = blockIdx.y;
    const int v = slot_v[slot], l = slot_l[slot];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        double2 t;
        t.x = V[(size_t)v * N + i] + cl[(size_t)l * N + i];
        t.y = uniform ? V[(size_t)v * N + i] : e2[i];
        tab[(size_t)slot * N + i] = t;
    }
}

// ---- boundary values on the device (GetMaxRadiusIndex / GetBoundaryValueFar, Numerov.h:103-136) -------
__device__ __forceinline__ double far_arg(const double* __restrict__ r, int i, double s, double delta)
{
    // Numerov.h:107:  -realPosition * sqrt(2|E|) - position * m_delta * 0.5
    return -r[i] * s - static_cast<double>(i) * delta * 0.5;
}

__global__ void k_boundary(const double* __restrict__ r, const double* __restrict__ E, int ntrials, GridScalars gs,
                           int* __restrict__ start, double* __restrict__ us, double* __restrict__ us1)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntrials) return;
    const double s = sqrt(2. * fabs(E[t]));
    int maxIndex = gs.N - 1, minIndex = 1;
    while (maxIndex - minIndex > 1) {   // Numerov.h:125-133 with exp(arg) < 1e-200 <=> arg < far_thr
        const int mid = (maxIndex + minIndex) / 2;
        if (far_arg(r, mid, s, gs.delta) < gs.far_thr) maxIndex = mid; else minIndex = mid;
    }
    start[t] = maxIndex;
    us[t] = exp(far_arg(r, maxIndex, s, gs.delta));
    us1[t] = exp(far_arg(r, maxIndex - 1, s, gs.delta));
}

// Uniform grid (Numerov.h:43-58, 274-296): the sweep starts at startPoint = min(Rmax, 200 / sqrt(2|E|)), at index
// (long)(startPoint / h); the two start values are exp(-position sqrt(2|E|)) at startPoint and startPoint - h.
__device__ __forceinline__ double uniform_start_point(double E, double Rmax)
{
    const double mr = 200. / sqrt(2. * fabs(E));      // GetMaxRadius, Numerov.h:53-56
    return mr < Rmax ? mr : Rmax;                      // std::min(startPoint, GetMaxRadius)
}

__global__ void k_boundary_uniform(const double* __restrict__ E, int ntrials, GridScalars gs, int* __restrict__ start,
                                   double* __restrict__ us, double* __restrict__ us1, int for_match,
                                   const int* __restrict__ l, double* __restrict__ uz)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntrials) return;
    const double s = sqrt(2. * fabs(E[t]));
    const double sp = uniform_start_point(E[t], gs.Rmax);
    const long steps = static_cast<long>(sp / gs.h);
    // the match solve re-derives the step from the truncated count (Numerov.h:430) before it steps back once
    const double hh = for_match ? sp / steps : gs.h;
    start[t] = static_cast<int>(steps);
    us[t] = exp(-sp * s);
    us1[t] = exp(-(sp - hh) * s);
    if (uz) uz[t] = pow(hh, static_cast<double>(l[t]) + 1.);       // GetBoundaryValueZero(h, l), Numerov.h:38-41
}

// ---- the sweep --------------------------------------------------------------------------------------------------
typedef unsigned long long lanemask_t;
constexpr int kChunk = 8;             // grid points per prefetched batch of the sweep body
constexpr int kBoundFrom = kChunk;    // the sweep body never touches i < kChunk (tail loop)
constexpr int kTinyFrom = 128;        // from this index outwards |f| is small enough for the series reciprocal where the slot's bound says so
constexpr double kTinyF = 12. / 2048.;   // |f| < 12 * 2^-11  <=>  |f/12| < 2^-11
constexpr int kPipeChunk = 32;        // grid points per stage of the pipelined kernel
constexpr int kPipeMaxBlocks = 768;   // above this many 64-trial blocks the fused kernel fills every SIMD anyway

struct SweepState {
    double w, wprev, u, fprev, prevSol;
};

// w / d for d in (0.5, 1.5) and |w| in [1e-270, 1e200] or 0: the instruction sequence hipcc emits for an IEEE fp64
// division (v_rcp_f64, two Newton steps on the reciprocal, quotient, remainder, final fma) minus v_div_scale /
// v_div_fmas' scaling / v_div_fixup, which are identities in this range -- so the result is bit-identical to `/`.
__device__ __forceinline__ double div_in_range(double w, double d)
{
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = w * r;
    const double rem = __builtin_fma(-d, q, w);
    return __builtin_fma(rem, r, q);
}

// The correctly rounded reciprocal of d = 1 - x for |x| < 2^-11 without v_rcp_f64 (a quarter-rate instruction: ~5 issue
// slots): r0 = 1 + x + x^2 + x^3 + x^4 is within max(|x|^5, 2^-52) < 2^-52 of 1/d, one Newton step squares that (pre-rounding
// error ~2^-104, the same as rcp + two steps), and reciprocals of 53-bit numbers stay further than that from a rounding
// boundary -- so r is RN(1/d), the value the compiler's division sequence arrives at, and the quotient w * r corrected by
// fma(-d, q, w) * r is the same IEEE quotient (tests compare sweeps bit for bit with the plain division).
__device__ __forceinline__ double recip_series(double x, double d)
{
    double t = __builtin_fma(x, x, x);
    t = __builtin_fma(x, t, x);
    t = __builtin_fma(x, t, x);
    const double r0 = 1.0 + t;
    const double e = __builtin_fma(-d, r0, 1.0);
    return __builtin_fma(r0, e, r0);
}

// One Numerov step for every lane of the wave (Numerov.h:311-321).  `tv` is wave-uniform (SGPRs).
// R2 = 2 * Rp2delta2: 2.*(veff - E)*Rp2delta2 == (veff - E)*(2*Rp2delta2) exactly (power-of-two scaling).
template <bool FAST>
__device__ __forceinline__ void numerov_step(SweepState& s, const double2 tv, const double E, const double R2, const double d2p4)
{
    const double wnext = 2. * s.w - s.wprev + s.u * s.fprev;     // Numerov.h:311 (h2 == 1)
    s.wprev = s.w;
    s.w = wnext;
    const double f = (tv.x - E) * R2 * tv.y + d2p4;               // Numerov.h:100
    const double d = 1. - kH2p12 * f;
    s.prevSol = s.u;
    s.u = FAST ? div_in_range(wnext, d) : wnext / d;              // getU, Numerov.h:510-513
    s.fprev = f;
}

// Bookkeeping of CountNodes (Numerov.h:323-340) for all 64 lanes at once.  The predicates are 64-bit lane masks in
// scalar registers (four v_cmp per point, the rest is scalar logic); the per-lane node budget (nodesLimit + 1 - nodes
// seen) is only touched in the rare wave-uniform branch "some lane crossed zero at this point".
struct CountState {
    lanemask_t live;      // still inside the reference's loop
    lanemask_t oldSgn;    // sign of the previous solution value
    lanemask_t flag;      // firstClassicalReturnPoint
};

// Returns the lanes that leave the loop at this point: bit set in .x because the classically allowed region ended
// (Numerov.h:338), in .y because the count exceeded the limit (Numerov.h:330).
struct CountExit { lanemask_t tp, over; };
__device__ __forceinline__ CountExit count_step(CountState& c, int& budget, const int lane, const double u, const double veff,
                                                 const double E, const lanemask_t started)
{
    const lanemask_t m_inf = __ballot(fabs(u) == INFINITY);
    const lanemask_t m_pos = __ballot(u > 0);
    const lanemask_t m_le = __ballot(veff <= E);
    const lanemask_t m_gt = __ballot(veff > E);           // separate compare: both false for NaN, as in the reference
    const lanemask_t act = c.live & started;
    const lanemask_t cross = act & ~m_inf & (m_pos ^ c.oldSgn);
    lanemask_t over = 0;
    if (cross != 0ull) {                                   // ++nodesCount; nodesCount > nodesLimit -> return
        budget -= (int)((cross >> lane) & 1ull);
        over = cross & __ballot(budget <= 0);
    }
    c.oldSgn = (c.oldSgn & ~act) | (m_pos & act);
    const lanemask_t stay = act & ~m_inf & ~over;
    const lanemask_t tp = stay & c.flag & m_gt;            // left the classically allowed region again
    c.flag |= stay & m_le;
    c.live = (c.live & ~started) | (stay & ~tp);
    return CountExit{tp, over};
}

struct SweepArgs {
    const double2* tab;        // slot tables
    const double2* bounds;     // per slot (stride bstride): { max_i |veff_i| R2 e2_i, max_i R2 e2_i } (fast-division range proof),
                               // followed by { min, max } of veff over the aligned blocks of kPipeChunk points; or null
    int bstride;
    const int* blk_slot;       // per block: table slot
    const int* blk_first;      // per block: first trial
    const int* blk_cnt;        // per block: number of trials (<= 64)
    const int* blk_kind;       // per block: DFTA_SWEEP_COUNT / DFTA_SWEEP_ZERO, or null -> `kind`
    int kind;
    const double* E;
    const int* limit;          // COUNT
    const int* start;
    const double* us;
    const double* us1;
    int* count;                // COUNT out
    double* u0;                // out (may be null for COUNT)
    int* trip;                 // out (may be null): loop iterations per trial (diagnostics)
    // COUNT, optional: where the sweep stopped and how far the nearest
    // zero of u is from that point.  istop = grid index of the last examined point (0: ran down to r = 0, -1: left for
    // another reason); phi = u_stop / (u_stop - u_prev), the position of the zero of the line through the last two values
    // in grid cells beyond the stop point -- a smooth, scale-free function of E that crosses 0 exactly where the counted
    // number of nodes changes.  The level solver interpolates it to predict the end point of a bisection (speculation only).
    double* phi;
    int* istop;                // | kStopOver: left because the count exceeded the limit, at the sign change that did it
    unsigned long long* total_trips;   // optional global counter (points traversed)
    const int* slot_l;         // uniform grid: l of every table slot
};

// Every wave owns up to 64 trials that share one table slot (potential, l).  The grid index loop is wave-uniform;
// table entries are fetched one chunk ahead as scalar loads (hidden behind the previous chunk's arithmetic).
// Every wave owns up to 64 trials that share one table slot (potential, l).  The grid index loop is wave-uniform.
// Table entries {veff_i, e2_i} are wave-uniform too; they are fetched with vector loads whose 64 lanes carry the same
// address (one 16-byte request per wave), CH points per batch and two batches ahead of the recurrence, so that the
// counted vmcnt waits hide the L2 latency.  Each batch is processed in two phases: first everything that does not
// depend on the recurrence (f, 1 - f/12 and its reciprocal: independent across the CH points), then the dependent
// chain w -> u -> w.
template <int KIND, int CH>
__device__ __forceinline__ void sweep_wave(const SweepArgs& a, const GridScalars& gs, const int b, const int lane)
{
    const int cnt = a.blk_cnt[b];
    const int t = a.blk_first[b] + (lane < cnt ? lane : 0);
    // a trial with start < 2 is a placeholder (unused slot of a bisection tree): the lane idles
    const bool valid = (lane < cnt) && (a.start[t] >= 2);
    const lanemask_t vmask = __ballot(valid);
    if (vmask == 0ull) return;       // whole wave idle
    const int slot = a.blk_slot[b];
    // per-lane view of the (wave-uniform) table row: a lane offset of zero keeps these on the vector-memory path
    const double2* __restrict__ T = a.tab + (size_t)slot * gs.N + __builtin_amdgcn_mbcnt_hi(0u, __builtin_amdgcn_mbcnt_lo(0u, 0u));
    const double R2 = 2. * gs.Rp2delta2;
    const double d2p4 = gs.delta2p4;

    const double E = a.E[t];
    const int start = valid ? a.start[t] : 2;
    const int limit = (KIND == DFTA_SWEEP_COUNT) ? a.limit[t] : 0;

    // can the whole wave use the division fast path?  |f| <= max|veff| R2 e2 + |E| max R2 e2 + delta^2/4 < 6 => d in (0.5, 1.5)
    // tiny: |f| < 12 * 2^-11 for every point the lane visits at or beyond kTinyFrom (e2 grows with i: the lane's own start
    // index bounds the |E| term) -- there the reciprocal comes from a short series instead of v_rcp_f64 (recip_series)
    bool fast = false, tiny = false;
    if (a.bounds) {
        const double2 bd = a.bounds[(size_t)slot * a.bstride];
        const bool lane_ok = !valid || (bd.x + fabs(E) * bd.y + d2p4 < 6.0);
        fast = (__ballot(lane_ok) == ~0ull);
        const double2 bt = a.bounds[(size_t)slot * a.bstride + a.bstride - 1];
        const bool lane_tiny = !valid || (bt.x + fabs(E) * (R2 * T[start].y) + d2p4 < kTinyF);
        tiny = fast && (__ballot(lane_tiny) == ~0ull);
    }

    // prologue (Numerov.h:293-306): the two far boundary points of this lane
    SweepState s;
    {
        const double2 ts = T[start];
        const double2 t1 = T[start - 1];
        const double us = a.us[t];
        s.u = a.us1[t];
        s.fprev = (ts.x - E) * R2 * ts.y + d2p4;
        s.wprev = (1 - kH2p12 * s.fprev) * us;
        s.fprev = (t1.x - E) * R2 * t1.y + d2p4;
        s.w = (1 - kH2p12 * s.fprev) * s.u;
        s.prevSol = us;
    }
    CountState c;
    c.live = vmask;
    c.oldSgn = __ballot(s.u > 0);
    c.flag = 0;
    int budget = limit + 1;              // nodes this lane may still count before nodesCount > nodesLimit
    const bool diag = (a.trip != nullptr);
    int trips = 0;                       // per lane, diagnostics only
    unsigned long long wave_trips = 0;   // wave-uniform

    const int my_hi = valid ? start - 2 : 0;   // this lane integrates i = my_hi .. 1
    int ihi = my_hi, ilo = valid ? my_hi : 0x7fffffff;
    for (int off = 32; off > 0; off >>= 1) {
        ihi = max(ihi, __shfl_xor(ihi, off));
        ilo = min(ilo, __shfl_xor(ilo, off));
    }
    ihi = __builtin_amdgcn_readfirstlane(ihi);
    ilo = __builtin_amdgcn_readfirstlane(ilo);   // from here down every valid lane has started

    // bookkeeping after a step; runs in wave-uniform control flow only, so that its lane masks stay in scalar registers
    // (uprev = u at the point above, idx = grid index: for SweepArgs::phi / istop)
    double phi = NAN;
    int istop = -1;
#define SWEEP_COUNT_U(uval, uprev, idx, veff, started)                                                 \
    if (KIND == DFTA_SWEEP_COUNT) {                                                                    \
        const lanemask_t before = c.live & (started);                                                  \
        if (diag) trips += (int)((before >> lane) & 1ull);                                             \
        wave_trips += __popcll(before);                                                                \
        const CountExit ex_ = count_step(c, budget, lane, (uval), (veff), E, (started));               \
        if ((ex_.tp | ex_.over) != 0ull && (((ex_.tp | ex_.over) >> lane) & 1ull)) {                   \
            phi = (uval) / ((uval) - (uprev));                                                         \
            istop = ((ex_.over >> lane) & 1ull) ? ((idx) | kStopOver) : (idx);                         \
        }                                                                                              \
        poisoned = true;                                                                               \
    }
#define SWEEP_COUNT(idx, veff, started) SWEEP_COUNT_U(s.u, s.prevSol, idx, veff, started)
    // Whole batches of the body are skipped by the bookkeeping when nothing can change in them (see the counter of the
    // pipelined kernel below: u keeps its sign and stays finite, veff stays on one side of E, for every live lane);
    // last_le = "veff <= E" at the last point that went through count_step, unusable while `poisoned`.
    lanemask_t last_le = 0;
    bool poisoned = true;

    int i = ihi;
    // head: lanes join one by one (per-step masking) until all valid lanes are in
    for (; i > ilo && i >= 1; --i) {
        const double2 tv = T[i];
        const bool in = valid && i <= my_hi;
        const lanemask_t started = __ballot(in);
        if (in) numerov_step<false>(s, tv, E, R2, d2p4);
        SWEEP_COUNT(i, tv.x, started)
    }
    // body: all valid lanes step together, no exec masking on the arithmetic.
    // Lanes that have left the reference's loop keep integrating (harmless), only their bookkeeping is frozen.
    if (i >= 2 * CH) {
        double2 A[CH], B[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) A[k] = T[i - k];
#pragma unroll
        for (int k = 0; k < CH; ++k) B[k] = T[i - CH - k];
        // one batch: CUR = T[i .. i-CH+1] and OTH = T[i-CH .. i-2CH+1] are both requested and i >= 2*CH.
        // Returns true when the body is finished (tail loop takes over at the new i).
        // The arithmetic of one batch in one of three division modes, chosen per batch (wave-uniform) so that the point
        // loop is straight-line code: 0 = IEEE division, 1 = the compiler's division sequence without its scaling steps
        // (rcp + two Newton steps, div_in_range), 2 = the same with the reciprocal from a series (recip_series).
        auto arith = [&](auto MODE_, double2 (&CUR)[CH], double (&uu)[CH], double (&veff)[CH]) {
            constexpr int MODE = decltype(MODE_)::value;
            double f[CH], d[CH], r[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                veff[k] = CUR[k].x;
                f[k] = (CUR[k].x - E) * R2 * CUR[k].y + d2p4;                // Numerov.h:100
                const double x = kH2p12 * f[k];
                d[k] = 1. - x;
                if (MODE == 2) r[k] = recip_series(x, d[k]);
                if (MODE == 1) {                                             // reciprocal of hipcc's fp64 division sequence
                    double rr = __builtin_amdgcn_rcp(d[k]);
                    double e = __builtin_fma(-d[k], rr, 1.0);
                    rr = __builtin_fma(rr, e, rr);
                    e = __builtin_fma(-d[k], rr, 1.0);
                    r[k] = __builtin_fma(rr, e, rr);
                }
            }
            // CUR is consumed: request the batch below OTH into it
            if (i - 2 * CH >= CH) {
#pragma unroll
                for (int k = 0; k < CH; ++k) CUR[k] = T[i - 2 * CH - k];
            }
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                // Numerov.h:311 (h2 == 1): 2 w is exact, so fma(2, w, -wprev) is the reference's 2 w - wprev
                const double wnext = __builtin_fma(2., s.w, -s.wprev) + s.u * s.fprev;
                s.wprev = s.w;
                s.w = wnext;
                s.prevSol = s.u;
                if (MODE == 0) s.u = wnext / d[k];                            // getU, Numerov.h:510-513
                else {
                    const double q = wnext * r[k];
                    const double rem = __builtin_fma(-d[k], q, wnext);
                    s.u = __builtin_fma(rem, r[k], q);
                }
                s.fprev = f[k];
                uu[k] = s.u;
            }
        };
        auto batch = [&](double2 (&CUR)[CH], double2 (&OTH)[CH]) -> bool {
            double uu[CH], veff[CH];
            const double ucarry = s.u;                                        // u at the point above this batch
            if (!fast) arith(std::integral_constant<int, 0>{}, CUR, uu, veff);
            else if (tiny && i - CH + 1 >= kTinyFrom) arith(std::integral_constant<int, 2>{}, CUR, uu, veff);
            else arith(std::integral_constant<int, 1>{}, CUR, uu, veff);
            if (KIND == DFTA_SWEEP_COUNT) {
                const lanemask_t act = c.live & vmask;
                bool quiet = false;
                if (!poisoned) {
                    double mn = uu[0], mx = uu[0], vmn = veff[0], vmx = veff[0];
#pragma unroll
                    for (int k = 1; k < CH; ++k) {
                        asm("v_min_f64 %0, %1, %2" : "=v"(mn) : "v"(mn), "v"(uu[k]));
                        asm("v_max_f64 %0, %1, %2" : "=v"(mx) : "v"(mx), "v"(uu[k]));
                        asm("v_min_f64 %0, %1, %2" : "=v"(vmn) : "v"(vmn), "v"(veff[k]));
                        asm("v_max_f64 %0, %1, %2" : "=v"(vmx) : "v"(vmx), "v"(veff[k]));
                    }
                    // an infinity or a NaN anywhere in the batch is still there at its last point: once w is not finite it
                    // never is again (2 w - wprev + u f keeps inf or turns it into NaN), so the last value speaks for all
                    const lanemask_t fin = __ballot(fabs(uu[CH - 1]) < INFINITY);
                    const lanemask_t allpos = __ballot(mn > 0), nonepos = __ballot(mx <= 0);
                    const lanemask_t le_all = __ballot(vmx <= E), gt_all = __ballot(vmn > E);
                    const lanemask_t ok = fin & ((c.oldSgn & allpos) | (~c.oldSgn & nonepos)) & ((le_all & last_le) | (gt_all & ~last_le));
                    quiet = (~ok & act) == 0ull;
                }
                if (quiet) {
                    if (diag) trips += CH * (int)((act >> lane) & 1ull);
                    wave_trips += (unsigned long long)CH * __popcll(act);
                } else {
#pragma unroll
                    for (int k = 0; k < CH; ++k) { SWEEP_COUNT_U(uu[k], (k == 0 ? ucarry : uu[k > 0 ? k - 1 : 0]), i - k, veff[k], vmask) }
                    const lanemask_t m_le = __ballot(veff[CH - 1] <= E), m_gt = __ballot(veff[CH - 1] > E);
                    last_le = m_le;
                    poisoned = ((m_le | m_gt) != ~0ull);
                }
            }
            if (fast) {
                // the fast path needs |w| in range: leave it for good as soon as any lane gets close to the edges
                const double au = fabs(s.u);
                const bool ok = !valid || (au < 1e200 && (au > 1e-270 || s.u == 0.0));
                fast = (__ballot(ok) == ~0ull);
            }
            i -= CH;
            if (KIND == DFTA_SWEEP_COUNT) {
                if ((c.live & vmask) == 0ull) { i = 0; return true; }
            }
            if (i < 2 * CH) {
                // OTH (already requested) is the last full batch: process it with the plain step and leave
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    numerov_step<false>(s, OTH[k], E, R2, d2p4);
                    SWEEP_COUNT(i - k, OTH[k].x, vmask)
                }
                i -= CH;
                return true;
            }
            return false;
        };
        while (true) {
            if (batch(A, B)) break;
            if (batch(B, A)) break;
        }
    }
    // tail
    for (; i >= 1; --i) {
        const double2 tv = T[i];
        numerov_step<false>(s, tv, E, R2, d2p4);
        SWEEP_COUNT(i, tv.x, vmask)
    }
#undef SWEEP_COUNT
#undef SWEEP_COUNT_U
    if (KIND == DFTA_SWEEP_ZERO) { trips = my_hi; wave_trips = 0; }

    const bool exited = valid && !((c.live >> lane) & 1ull);
    double u0 = NAN;
    int count = (KIND == DFTA_SWEEP_COUNT) ? limit + 1 - budget : 0;
    if (valid && (KIND == DFTA_SWEEP_ZERO || !exited)) {
        u0 = s.u * (2 + s.fprev) - s.prevSol;                 // Numerov.h:345 / 398
        if (KIND == DFTA_SWEEP_COUNT) {
            const bool oldSgn = (c.oldSgn >> lane) & 1ull;
            if ((u0 > 0) != oldSgn) ++count;                  // Numerov.h:346-347
            phi = u0 / (u0 - s.u);
            istop = 0;
        }
    }
    if (valid) {
        if (KIND == DFTA_SWEEP_COUNT) a.count[t] = count;
        if (a.u0) a.u0[t] = u0;
        if (a.trip) a.trip[t] = trips;
        if (KIND == DFTA_SWEEP_COUNT && a.phi) a.phi[t] = phi;
        if (KIND == DFTA_SWEEP_COUNT && a.istop) a.istop[t] = istop;
    }
    if (a.total_trips) {
        if (KIND == DFTA_SWEEP_ZERO) {
            int sum = valid ? trips : 0;
            for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
            wave_trips = (unsigned long long)sum;
        }
        if (lane == 0) atomicAdd(a.total_trips, wave_trips);
    }
}

template <int CH>
__global__ __launch_bounds__(256) void k_sweep(SweepArgs a, GridScalars gs, int nwaves)
{
    const int lane = threadIdx.x & 63;
    const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (b >= nwaves) return;
    const int kind = a.blk_kind ? __builtin_amdgcn_readfirstlane(a.blk_kind[b]) : a.kind;
    if (kind == DFTA_SWEEP_COUNT) sweep_wave<DFTA_SWEEP_COUNT, CH>(a, gs, b, lane);
    else                          sweep_wave<DFTA_SWEEP_ZERO, CH>(a, gs, b, lane);
}

// The same sweeps launched LONGEST BLOCK FIRST (batches: levels.hip).  A launch of k_sweep hands its blocks out in the order of the trial
// arrays, all at once when they fit (two waves per SIMD), and ends when the SIMD that happened to get two full-length blocks ends: 131 k
// points x 2 x 24 instructions at the SIMD's issue rate, while the average block runs 53 k points (an over-limit CountNodes block leaves
// after a few thousand, an l > 0 one stops at its inner turning point) -- the round-5 batch sat at 0.29 of the fp64 issue ceiling.
// Here k_expand has entered the blocks that have work into kSweepQueueClasses lists by expected length; workgroup t (ONE wave: its
// registers go back to the dispatcher the moment its block ends) takes entry t of the lists laid end to end, longest class first.  The
// dispatcher starts workgroups in index order, so the long blocks start first and the short ones fill the SIMDs as they come free:
// longest-processing-time-first -- the launch ends within a short block of max(total work / SIMDs, one full-length block).
template <int CH>
__global__ __launch_bounds__(64) void k_sweep_queue(SweepArgs a, GridScalars gs, const int* __restrict__ qcnt, const int* __restrict__ qlist, int qcap)
{
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x;
    int cls = 0, base = 0;
    for (; cls < kSweepQueueClasses; ++cls) {
        const int n = qcnt[cls];
        if (t < base + n) break;
        base += n;
    }
    if (cls == kSweepQueueClasses) return;            // fewer blocks with work than the round has blocks
    const int b = __builtin_amdgcn_readfirstlane(qlist[(size_t)cls * qcap + (t - base)]);
    const int kind = a.blk_kind ? __builtin_amdgcn_readfirstlane(a.blk_kind[b]) : a.kind;
    if (kind == DFTA_SWEEP_COUNT) sweep_wave<DFTA_SWEEP_COUNT, CH>(a, gs, b, lane);
    else                          sweep_wave<DFTA_SWEEP_ZERO, CH>(a, gs, b, lane);
}

// ---- pipelined sweep: one workgroup = one block of 64 trials, its five waves are the stages of a pipeline ----------
// A SIMD of gfx950 issues one fp64 VALU instruction of a wave64 every 4 cycles, whether it has one wave or several.
// The fused kernel above spends ~19 of them per grid point in ONE wave (plus exposed v_rcp_f64 / load latency), i.e.
// 80-90 ns per point and trial block when the machine is not full -- the situation of a single atom (a few hundred
// blocks on 1024 SIMDs).  Only 5 of those instructions form the loop-carried chain u -> u f -> w -> q -> rem -> u.
// Here the work of a block is spread over the four SIMDs of a compute unit (wave w runs on SIMD w mod 4):
//     waves 0,1,3  producers   f_i and the refined reciprocal r_i of d_i = 1 - f_i/12, 1/4 + 3/8 + 3/8 of a chunk  -> LDS
//     wave  2      integrator  the loop-carried recurrence (8 VALU instructions per point, d_i recomputed); per chunk
//                              it leaves w, wprev, u after the last point and the sign bits of u  -> LDS
//     wave  4      counter     CountNodes' bookkeeping: a chunk in which veff stays on its side of E is settled from the
//                              sign bits (crossings = sign changes); otherwise the counter integrates the chunk again
//                              itself and replays it point by point with lane masks in scalar registers; shares
//                              SIMD 0 with a producer
// The stages are chunks of CH grid points apart (software pipeline, one s_barrier per chunk): at iteration `it` the
// producers write chunk it, the integrator integrates chunk it-2 (reading the rest of it and the beginning of chunk
// it-1 into a rolling register window), the counter examines chunk it-3.  Every floating-point operation is the one the
// fused kernel executes, in the same order, so results are bit-identical.
typedef double v2d __attribute__((ext_vector_type(2)));

template <int CH>
struct PipeShared {
    v2d rows[4][CH];             // table rows {veff_i, e2_i} of the last four chunks (wave-uniform data)
    double prod[4][CH][2][64];   // f_i and the refined reciprocal r_i of d_i = 1 - f_i/12, per point and lane
    double st[4][4][64];         // integrator -> counter, per chunk and lane: w, wprev, u after the chunk's last point and a
                                 // summary of the signs of u over the chunk (two int32 in one double, see sweep_pipe)
    double fin[64];              // u(0) of every lane at the end of a COUNT sweep
    double fin1[64];             // and u at grid point 1
    int stop;                    // set by the counter when every lane has left CountNodes' loop
};

// One table row per lane, issued without the compiler's bookkeeping ...
__device__ __forceinline__ void tab_load(v2d& x, const double2* p)
{
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(x) : "v"(p) : "memory");
}
// ... and the matching wait: vector-memory loads return in order, so "at most YOUNGER loads outstanding" means X has
// landed.  X is an in/out operand so that no use of it can be scheduled above the wait.  (hipcc's own waitcnt
// insertion drains vmcnt to 0 at a loop header, which would expose the full memory latency once per trip.)
template <int YOUNGER>
__device__ __forceinline__ void tab_wait(v2d& X)
{
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(X) : "n"(YOUNGER) : "memory");
}

__device__ __forceinline__ double read_lane(double v, int k)
{
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)b, k), hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), k);
    return __builtin_bit_cast(double, ((long long)hi << 32) | lo);
}

__device__ __forceinline__ int hi_word(double v) { return (int)(__builtin_bit_cast(long long, v) >> 32); }
constexpr int kSignWord = 0x7fffdead;   // second word of a chunk summary that holds sign bits (the high word of a NaN no arithmetic produces)

// s_barrier without the vmcnt(0) that __syncthreads() implies: table prefetches stay in flight across the barrier
#ifdef DFTA_PIPE_PROF
// measurements only: per role { ticks waiting at the barrier (LDS drain included), barriers } of block 0
__device__ unsigned long long g_pipe_prof[8 * 2];
#define PIPE_BARRIER() do { const unsigned long long t0_ = __builtin_readcyclecounter(); \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) { g_pipe_prof[(threadIdx.x >> 6) * 2] += __builtin_readcyclecounter() - t0_; g_pipe_prof[(threadIdx.x >> 6) * 2 + 1] += 1; } } while (0)
#else
#define PIPE_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

struct PipeTrial {          // what every wave of the block knows about its lane's trial
    double E;
    int my_hi;              // this lane integrates i = my_hi .. 1
    bool valid;
    int ihi, ilo;           // wave-uniform: max / min of my_hi over the valid lanes
    int nch, nit, N;
};

// Producer of the points OFF .. OFF+CNT-1 of every chunk.  Lane k < CNT fetches the table row of point OFF+k, kPD = 4
// chunks ahead (one vector load per chunk and producer); when the row set has landed it is parked in LDS, from where
// every lane reads the rows back as broadcasts -- wave-uniform operands at the cost of LDS instructions, not VALU ones.
template <bool COUNT, int CH, int OFF, int CNT>
__device__ __forceinline__ bool pipe_producer(PipeShared<CH>& sh, const double2* __restrict__ T, const PipeTrial& tr, const int lane,
                                              const double R2, const double d2p4)
{
    const double E = tr.E;
    v2d X0, X1, X2, X3;
    auto load = [&](v2d& X, int c) {
        const int i = (tr.nch - c) * CH - OFF - lane;       // rows above ihi (first chunk) and below 1 (past the end) are never used
        if (lane < CNT) tab_load(X, T + min(max(i, 1), tr.N - 1));
    };
    load(X0, 0);
    load(X1, 1);
    load(X2, 2);
    load(X3, 3);
    int ps = 0;                                            // it % 4
    volatile int* stop = &sh.stop;
    auto stage = [&](v2d& X, int it) -> bool {
        tab_wait<3>(X);                                    // the load of X is older than the three behind it
        // unconditional (chunks past the end clamp at i = 1 and are never read): the wait counts stay exact
        if (COUNT && lane < CNT) sh.rows[it & 3][OFF + lane] = X;           // veff_i for the counter's point-by-point path
        double* P = &sh.prod[ps][OFF][0][lane];
#pragma unroll
        for (int kk = 0; kk < CNT; ++kk) {
            const double veff = read_lane(X.x, kk), e2 = read_lane(X.y, kk);   // wave-uniform operands (scalar registers)
            const double f = (veff - E) * R2 * e2 + d2p4;                    // Numerov.h:100
            const double d = 1. - kH2p12 * f;
            double rr = __builtin_amdgcn_rcp(d);                             // reciprocal of hipcc's fp64 division sequence
            double e = __builtin_fma(-d, rr, 1.0);
            rr = __builtin_fma(rr, e, rr);
            e = __builtin_fma(-d, rr, 1.0);
            rr = __builtin_fma(rr, e, rr);
            P[kk * 128] = f;
            P[kk * 128 + 64] = rr;
        }
        load(X, it + 4);
        ps = (ps + 1) & 3;
        PIPE_BARRIER();
        return COUNT && (it & 7) == 7 && *stop != 0;
    };
    bool stopped = false;
    for (int it = 0; it < tr.nit; it += 4) {
        if (stage(X0, it)) { stopped = true; break; }
        if (stage(X1, it + 1)) { stopped = true; break; }
        if (stage(X2, it + 2)) { stopped = true; break; }
        if (stage(X3, it + 3)) { stopped = true; break; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // retire the prefetches that ran past the end
    return stopped;
}

template <int KIND, int CH>
__device__ __forceinline__ void sweep_pipe(const SweepArgs& a, const GridScalars& gs, const int b, const int lane, const int role,
                                           PipeShared<CH>& sh)
{
    constexpr bool COUNT = (KIND == DFTA_SWEEP_COUNT);
    const int cnt = a.blk_cnt[b];
    const int t = a.blk_first[b] + (lane < cnt ? lane : 0);
    const bool valid = (lane < cnt) && (a.start[t] >= 2);
    const lanemask_t vmask = __ballot(valid);
    if (vmask == 0ull) return;       // same decision in all waves
    const int slot = a.blk_slot[b];
    const double2* __restrict__ T = a.tab + (size_t)slot * gs.N;
    const double R2 = 2. * gs.Rp2delta2;
    const double d2p4 = gs.delta2p4;
    const double E = a.E[t];
    const int start = valid ? a.start[t] : 2;
    const int limit = COUNT ? a.limit[t] : 0;

    PipeTrial tr;
    tr.E = E;
    tr.N = gs.N;
    tr.valid = valid;
    tr.my_hi = valid ? start - 2 : 0;
    const int my_hi = tr.my_hi;
    int ihi = my_hi, ilo = valid ? my_hi : 0x7fffffff;
    for (int off = 32; off > 0; off >>= 1) {
        ihi = max(ihi, __shfl_xor(ihi, off));
        ilo = min(ilo, __shfl_xor(ilo, off));
    }
    ihi = tr.ihi = __builtin_amdgcn_readfirstlane(ihi);
    ilo = tr.ilo = __builtin_amdgcn_readfirstlane(ilo);
    const int nch = tr.nch = (ihi + CH - 1) / CH;          // chunk c is the aligned block i = (nch - c) CH .. (nch - c - 1) CH + 1
    const int nit = tr.nit = (nch + (COUNT ? 3 : 2) + 3) & ~3;   // multiple of four: the stage loops are unrolled by two or four
    volatile int* stop = &sh.stop;
    if (threadIdx.x == 0) *stop = 0;
    PIPE_BARRIER();
    bool stopped = false;
    // a chunk can take the straight-line paths when it is complete, inside the range of the division bounds, and
    // no lane joins the sweep inside it (every lane is in or out for the whole chunk)
    auto plain_chunk = [&](int top) -> bool {
        if (top == CH) return false;                       // the innermost block holds i < kBoundFrom
        if (top < ilo) return true;                        // everybody joined earlier
        const bool joins = valid && my_hi <= top && my_hi > top - CH;   // (a lane that starts at `top` itself must be reset too)
        return __ballot(joins) == 0ull;
    };

    // the state a lane starts from at i = my_hi (integrator; counter when it integrates a chunk again)
    auto initial_state = [&]() -> SweepState {
        SweepState s;
        const double2 ts = T[start];
        const double2 t1 = T[start - 1];
        const double us = a.us[t];
        s.u = a.us1[t];
        s.fprev = (ts.x - E) * R2 * ts.y + d2p4;
        s.wprev = (1 - kH2p12 * s.fprev) * us;
        s.fprev = (t1.x - E) * R2 * t1.y + d2p4;
        s.w = (1 - kH2p12 * s.fprev) * s.u;
        s.prevSol = us;
        return s;
    };

    // the producer on SIMD 0 shares it with the counter: it takes a quarter of the chunk, the other two 3/8 each
#ifndef DFTA_PIPE_P0
#define DFTA_PIPE_P0 (CH / 4)
#endif
    constexpr int kP0 = DFTA_PIPE_P0, kP1 = (CH - kP0) / 2;
    if (role == 0)      stopped = pipe_producer<COUNT, CH, 0, kP0>(sh, T, tr, lane, R2, d2p4);
    else if (role == 1) stopped = pipe_producer<COUNT, CH, kP0, kP1>(sh, T, tr, lane, R2, d2p4);
    else if (role == 3) stopped = pipe_producer<COUNT, CH, kP0 + kP1, CH - kP0 - kP1>(sh, T, tr, lane, R2, d2p4);
    else if (role == 2) {
        // ---------------- integrator
        bool fast = false;
        if (a.bounds) {
            const double2 bd = a.bounds[(size_t)slot * a.bstride];
            const bool lane_ok = !valid || (bd.x + fabs(E) * bd.y + d2p4 < 6.0);
            fast = (__ballot(lane_ok) == ~0ull);
        }
        SweepState s = initial_state();
        const SweepState s0 = s;                               // a lane (re)starts from here at i = my_hi
        // {f, r} of the points ahead, in registers: entry e of a chunk lives in R.f[e], R.r[e]; when a stage begins the first
        // kAhead entries of its chunk are there, the others follow while the chunk is integrated -- and then the first kAhead
        // of the next chunk (whose buffer the producers completed a stage ago) -- four reads per three points, so that the
        // last one is issued a quarter of a chunk before the barrier (which waits for the wave's outstanding LDS operations).
        // Only the entries between "read" and "used" are live: ~kAhead + CH/4 of them.
        constexpr int kAhead = CH / 2, kRd = CH - CH / 4;
        static_assert(kRd * 4 == CH * 3 && kAhead * 2 == CH, "chunk size must be a multiple of four");
        struct Regs { double f[CH], r[CH]; };
        Regs R;
        auto stage = [&](int it) -> bool {
            // chunk it-2 is integrated (reads of a chunk that does not exist return stale LDS contents that are never used)
            const int c = it - 2;
            const double* Pc = &sh.prod[c & 3][0][0][lane];
            const double* Pn = &sh.prod[(c + 1) & 3][0][0][lane];
            auto fetch = [&](int q) {                          // entry q + kAhead of the running sequence (q: compile time)
                const int e = q + kAhead;
                if (e < CH) { R.f[e] = Pc[e * 128]; R.r[e] = Pc[e * 128 + 64]; }
                else        { R.f[e - CH] = Pn[(e - CH) * 128]; R.r[e - CH] = Pn[(e - CH) * 128 + 64]; }
            };
            const bool have = c >= 0 && c < nch;
            const int top = (nch - c) * CH;
            // What the counter gets of a chunk (COUNT): not the CH values of u (a 64-lane ds_write costs the integrator
            // 6 .. 18 ns), but a summary of their signs and w, wprev, u after the last point, from which the counter
            // integrates the chunk itself in the rare case that it has to look at every point.  Summary of a chunk from the
            // straight-line path: the CH sign bits of u, first point in the top bit (one v_alignbit_b32 per point; second word
            // kSignWord).  There d is in (0.5, 1.5) and |w| >= 2^-969 or 0, so u = +-0 only where w = 0, and then
            // w_next = -w_prev: a zero inside a run of values is always followed by the opposite sign (or by zeros to the
            // end of the sweep) -- as long as the chunk's last value is finite and not zero, "u > 0" is "sign bit clear" at
            // every point where it matters and the sign changes of the word are CountNodes' crossings.  Summary from the
            // general path: min and max of the HIGH WORDS as signed integers (min > 0: every u > 0; max < 0: every sign
            // bit set; +0 counts as "cannot tell").
            int hmin = 0x7fffffff, hmax = (int)0x80000000;
            unsigned sw = 0;
            bool straight = false;
            if (have && fast && plain_chunk(top)) {
                straight = true;
                // Straight-line code for all 64 lanes: lanes that have not started yet integrate garbage (they are reset
                // to s0 when they join, in the other branch).  The LDS reads are interleaved with the recurrence (8 VALU,
                // then 1 or 2 DS reads per point) -- issued in one burst they would stall the wave.
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const double fk = R.f[k], rk = R.r[k];
                    // Numerov.h:311 (h2 == 1): 2 w is exact, so fma(2, w, -wprev) is the reference's 2 w - wprev
                    const double wnext = __builtin_fma(2., s.w, -s.wprev) + s.u * s.fprev;
                    s.wprev = s.w;
                    s.w = wnext;
                    s.prevSol = s.u;
                    const double d = 1. - kH2p12 * fk;                       // off the loop-carried chain
                    const double q = wnext * rk;
                    const double rem = __builtin_fma(-d, q, wnext);
                    s.u = __builtin_fma(rem, rk, q);
                    s.fprev = fk;
                    if (COUNT) sw = __builtin_amdgcn_alignbit(sw, (unsigned)hi_word(s.u), 31);   // (sw << 1) | sign bit
                    if (k < kRd) {
                        const int q0 = (4 * k) / 3, q1 = (4 * (k + 1)) / 3;
                        fetch(q0);
                        if (q1 - q0 == 2) fetch(q0 + 1);
                    }
                    if (COUNT) __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
                    else       __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                    if (k < kRd) {
                        if ((4 * (k + 1)) / 3 - (4 * k) / 3 == 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        else                                      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < CH - kAhead; ++q) fetch(q);              // the rest of this chunk
                __builtin_amdgcn_sched_barrier(0);
                if (have) {
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        const int i = top - k;
                        const bool fdiv = fast && i >= kBoundFrom;
                        if (valid && i <= my_hi) {
                            if (i == my_hi) s = s0;
                            const double wnext = 2. * s.w - s.wprev + s.u * s.fprev;
                            s.wprev = s.w;
                            s.w = wnext;
                            s.prevSol = s.u;
                            const double d = 1. - kH2p12 * R.f[k];
                            if (fdiv) {
                                const double q = wnext * R.r[k];
                                const double rem = __builtin_fma(-d, q, wnext);
                                s.u = __builtin_fma(rem, R.r[k], q);
                            } else {
                                s.u = wnext / d;                                          // getU, Numerov.h:510-513
                            }
                            s.fprev = R.f[k];
                        }
                        if (COUNT) { const int h = hi_word(s.u); hmin = min(hmin, h); hmax = max(hmax, h == kSignWord ? 0x7ff80000 : h); }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = CH - kAhead; q < CH; ++q) fetch(q);             // the beginning of the next one
            }
            if (COUNT && have) {
                double* ST = &sh.st[c & 3][0][lane];
                ST[0] = s.w;
                ST[64] = s.wprev;
                ST[128] = s.u;
                ST[192] = straight ? __builtin_bit_cast(double, ((long long)kSignWord << 32) | sw)
                                   : __builtin_bit_cast(double, ((long long)hmax << 32) | (unsigned)hmin);
            }
            if (have && fast) {
                // the reciprocal path needs |w| in [2^-969, 2^767) -- outside, v_div_scale rescales the operands -- all the way
                // through the next chunk: leave it for good when any started lane gets within 1.6 decades per step of a chunk
                // (a generous bound for |f| < 6) of the edges, plus 16 decades for a value next to a node at the lower one
                static_assert(CH <= 32, "range margins of the reciprocal path");
                constexpr double kLo = CH <= 16 ? 1e-250 : CH <= 24 ? 1e-237 : 1e-224;
                constexpr double kHi = CH <= 16 ? 1e200 : CH <= 24 ? 1e192 : 1e179;
                const double au = fabs(s.u);
                const bool ok = !(valid && my_hi > top - CH) || (au < kHi && (au > kLo || s.u == 0.0));
                fast = (__ballot(ok) == ~0ull);
            }
            PIPE_BARRIER();
            return COUNT && (it & 7) == 7 && *stop != 0;
        };
        for (int it = 0; it < nit; ++it)
            if (stage(it)) { stopped = true; break; }
        if (my_hi == 0) s = s0;                                // start == 2: nothing to integrate, u(0) comes from the boundary values
        if (!COUNT) {
            if (valid) {
                if (a.u0) a.u0[t] = s.u * (2 + s.fprev) - s.prevSol;                 // Numerov.h:398
                if (a.trip) a.trip[t] = my_hi;
            }
            if (a.total_trips) {
                int sum = valid ? my_hi : 0;
                for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
                if (lane == 0) atomicAdd(a.total_trips, (unsigned long long)sum);
            }
        } else if (!stopped) {
            sh.fin[lane] = s.u * (2 + s.fprev) - s.prevSol;                          // Numerov.h:345
            sh.fin1[lane] = s.u;
        }
    } else {
        // ---------------- counter
        CountState c;
        c.live = vmask;
        c.oldSgn = __ballot(a.us1[t] > 0);
        c.flag = 0;
        int budget = limit + 1;
        const bool diag = (a.trip != nullptr);
        int trips = 0;
        unsigned long long wave_trips = 0;
        // { min, max } of veff per aligned block of CH points (k_block_minmax)
        const double2* __restrict__ blkmm =
            a.bounds ? a.bounds + (size_t)slot * a.bstride + 1 : nullptr;
        // 64 consecutive entries at a time, one per lane (block bhi - lane), the next 64 already on their way; the entry of
        // a chunk is broadcast from its lane
        auto mm_fetch = [&](int bhi) -> double2 { return blkmm ? blkmm[max(bhi - lane, 0)] : double2{0., 0.}; };
        double2 mmv = mm_fetch(nch - 1), mmv_next = mm_fetch(nch - 1 - 64);
        lanemask_t last_le = 0;       // veff <= E at the last point that went through count_step
        bool poisoned = true;         // last_le unusable (nothing processed yet, or a NaN veff)
        double phi = NAN, ucarry = NAN;   // see SweepArgs::phi; u at the last point of the previous chunk
        double fcarry = 0;                // f at that point
        int istop = -1;
        const SweepState s0 = COUNT ? initial_state() : SweepState{};
        for (int it = 0; it < nit; ++it) {
            const int cc = it - 3;
            if (COUNT && cc >= 0 && cc < nch) {
                const int top = (nch - cc) * CH;
                const double* ST = &sh.st[cc & 3][0][lane];
                const double* F = &sh.prod[cc & 3][0][0][lane];
                const v2d* rows = &sh.rows[cc & 3][0];
                const double ulast = ST[128];
                const long long hmm = __builtin_bit_cast(long long, ST[192]);
                const int hmin = (int)hmm, hmax = (int)(hmm >> 32);
                const bool signs = (__builtin_amdgcn_readfirstlane(hmax) == kSignWord);   // wave-uniform, as the integrator's path is
                const unsigned sw = (unsigned)hmin;
                const double flast = F[(CH - 1) * 128];
                const double2 mm = double2{read_lane(mmv.x, cc & 63), read_lane(mmv.y, cc & 63)};   // block nch - 1 - cc
                if ((cc & 63) == 63) { mmv = mmv_next; mmv_next = mm_fetch(nch - 1 - (cc + 1) - 64); }
                bool quiet = false;
                const lanemask_t started = __ballot(valid && my_hi >= top);
                const lanemask_t act = c.live & started;
                if (!poisoned && blkmm && plain_chunk(top)) {
                    // Nothing can change in this chunk if, for every active lane, veff stays on the side of E it was on
                    // at the last examined point, u keeps its sign and stays finite: then cross = tp = 0 and flag, live,
                    // oldSgn are fixed points of count_step (flag already holds stay & m_le, live already lost flag & m_gt).
                    // Per lane, from the integrator: u after the chunk's last point (an infinity or a NaN survives to there
                    // through the recurrence, and a NaN veff makes u NaN) and min / max of the high words of u over the
                    // chunk: min > 0 -- every u positive; max < 0 -- every sign bit set.  (+0 and the denormals below
                    // 2^-1042 count as "cannot tell": the chunk is then examined point by point.)
                    const lanemask_t fin = __ballot(fabs(ulast) < INFINITY);
                    const lanemask_t le_all = __ballot(mm.y <= E), gt_all = __ballot(mm.x > E);
                    const lanemask_t v_ok = (le_all & last_le) | (gt_all & ~last_le);
                    lanemask_t allpos, nonepos;
                    if (signs) {
                        // Sign bits of every point.  With veff on its side and everything finite, count_step only counts
                        // crossings and moves oldSgn: x has a bit for every point whose sign differs from the point before
                        // (the first one from oldSgn), so the chunk is settled here unless a lane runs out of budget -- then
                        // the exit point is needed -- or ends on a zero.
                        const unsigned mask = CH >= 32 ? 0xffffffffu : ((1u << (CH & 31)) - 1u);
                        const unsigned prevbit = ((c.oldSgn >> lane) & 1ull) ? 0u : 1u;
                        const unsigned x = (sw ^ ((sw >> 1) | (prevbit << (CH - 1)))) & mask;
                        const int crosses = __popc(x);
                        const bool lane_act = (act >> lane) & 1ull;
                        const lanemask_t settled = fin & v_ok & __ballot(ulast != 0.0) & __ballot(budget - crosses > 0);
                        if ((~settled & act) == 0ull) {
                            quiet = true;
                            if (lane_act) budget -= crosses;
                            const lanemask_t lastpos = __ballot((sw & 1u) == 0u);
                            c.oldSgn = (c.oldSgn & ~act) | (lastpos & act);
                        }
                        allpos = nonepos = 0;
                    } else {
                        allpos = __ballot(hmin > 0);
                        nonepos = __ballot(hmax < 0);
                        const lanemask_t ok = fin & ((c.oldSgn & allpos) | (~c.oldSgn & nonepos)) & v_ok;
                        quiet = (~ok & act) == 0ull;
                    }
#ifdef DFTA_PIPE_PROF
                    if (lane == 0) {
                        const lanemask_t sgn_ok = fin & ((c.oldSgn & allpos) | (~c.oldSgn & nonepos));
                        atomicAdd(&g_pipe_prof[10], 1ull);                                       // plain chunks examined
                        if (!quiet) atomicAdd(&g_pipe_prof[(~v_ok & act) ? 12 : 11], 1ull);      // 11: sign / finiteness only, 12: veff side
                        (void)sgn_ok;
                    }
                } else if (lane == 0) {
                    atomicAdd(&g_pipe_prof[13], 1ull);                                           // joins, innermost chunk, poisoned
#endif
                }
                if (quiet) {
                    if (diag) trips += CH * (int)((act >> lane) & 1ull);
                    wave_trips += (unsigned long long)CH * __popcll(act);
                } else {
                    // The chunk again, point by point: u is integrated once more from the state the integrator left after
                    // the chunk before (IEEE division: bit for bit what the integrator's reciprocal path returns).
                    SweepState s = s0;
                    if (cc > 0) {
                        const double* SP = &sh.st[(cc - 1) & 3][0][lane];
                        s.w = SP[0];
                        s.wprev = SP[64];
                        s.u = SP[128];
                        s.fprev = fcarry;
                    }
                    double u[CH];
#pragma unroll
                    for (int k = 0; k < CH; ++k) u[k] = F[k * 128];            // f_i first, overwritten by u_i below
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        const int i = top - k;
                        const double f = u[k];
                        if (valid && i <= my_hi) {
                            if (i == my_hi) s = s0;
                            const double wnext = 2. * s.w - s.wprev + s.u * s.fprev;
                            s.wprev = s.w;
                            s.w = wnext;
                            const double d = 1. - kH2p12 * f;
                            s.u = wnext / d;
                            s.fprev = f;
                        }
                        u[k] = s.u;
                    }
                    lanemask_t m_le = 0, m_gt = 0;
#pragma unroll
                    for (int k = 0; k < CH; ++k) {
                        const int i = top - k;
                        if (i >= 1) {
                            const lanemask_t st = __ballot(valid && i <= my_hi);
                            const lanemask_t before = c.live & st;
                            if (diag) trips += (int)((before >> lane) & 1ull);
                            wave_trips += __popcll(before);
                            const double veff = rows[k].x;
                            const CountExit ex = count_step(c, budget, lane, u[k], veff, E, st);
                            if ((ex.tp | ex.over) != 0ull && (((ex.tp | ex.over) >> lane) & 1ull)) {
                                const double up = (k == 0) ? ucarry : u[k > 0 ? k - 1 : 0];
                                phi = u[k] / (u[k] - up);
                                istop = ((ex.over >> lane) & 1ull) ? (i | kStopOver) : i;
                            }
                            m_le = __ballot(veff <= E);
                            m_gt = __ballot(veff > E);
                        }
                    }
                    last_le = m_le;
                    poisoned = ((m_le | m_gt) != ~0ull);
                    if ((c.live & vmask) == 0ull && lane == 0) *stop = 1;
                }
                ucarry = ulast;
                fcarry = flast;
            }
            PIPE_BARRIER();
            if (COUNT && (it & 7) == 7 && *stop != 0) { stopped = true; break; }
        }
        if (COUNT) {
            if (!stopped) PIPE_BARRIER();          // pairs with the barrier below: sh.fin is complete
            const bool exited = valid && !((c.live >> lane) & 1ull);
            double u0 = NAN;
            int count = limit + 1 - budget;
            if (valid && !exited) {
                u0 = sh.fin[lane];
                const bool oldSgn = (c.oldSgn >> lane) & 1ull;
                if ((u0 > 0) != oldSgn) ++count;                                      // Numerov.h:346-347
                phi = u0 / (u0 - sh.fin1[lane]);
                istop = 0;
            }
            if (valid) {
                a.count[t] = count;
                if (a.u0) a.u0[t] = u0;
                if (a.trip) a.trip[t] = trips;
                if (a.phi) a.phi[t] = phi;
                if (a.istop) a.istop[t] = istop;
            }
            if (a.total_trips && lane == 0) atomicAdd(a.total_trips, wave_trips);
        }
        return;
    }
    if (COUNT && !stopped) PIPE_BARRIER();   // producers and integrator: sh.fin handed to the counter
}

constexpr int kPipeThreads = 320;

template <int CH>
__global__ __launch_bounds__(kPipeThreads) void k_sweep_pipe(SweepArgs a, GridScalars gs, int nblocks, const int* __restrict__ qcnt, const int* __restrict__ qlist, int qcap)
{
    __shared__ PipeShared<CH> sh;
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int b = blockIdx.x;
    if (qcnt) {
        // longest block first (see k_sweep_queue): with more blocks than compute units the second pass is made of the short ones
        int cls = 0, base = 0;
        for (; cls < kSweepQueueClasses; ++cls) {
            const int n = qcnt[cls];
            if (b < base + n) break;
            base += n;
        }
        if (cls == kSweepQueueClasses) return;
        b = __builtin_amdgcn_readfirstlane(qlist[(size_t)cls * qcap + (b - base)]);
    }
    const int kind = a.blk_kind ? __builtin_amdgcn_readfirstlane(a.blk_kind[b]) : a.kind;
    if (kind == DFTA_SWEEP_COUNT) sweep_pipe<DFTA_SWEEP_COUNT, CH>(a, gs, b, lane, role, sh);
    else                          sweep_pipe<DFTA_SWEEP_ZERO, CH>(a, gs, b, lane, role, sh);
}

// per slot: { max_i |veff_i| R2 e2_i, max_i R2 e2_i } over i = kBoundFrom .. N-1 (NaN poisons the bound -> slow division).
// The innermost points are excluded: there f ~ l(l+1)/i^2 (f_1 = 12 for l = 3, i.e. 1 - f/12 = 0), and they are always
// integrated by the tail loop with the plain IEEE division.
constexpr int kBoundsThreads = 1024;   // one block per slot: the loop is a chain of memory round trips, so as many loads in flight as a block can hold
__global__ __launch_bounds__(kBoundsThreads) void k_slot_bounds(const double2* __restrict__ tab, int N, double R2, double2* __restrict__ bounds, int bstride)
{
    constexpr int kW = kBoundsThreads / 64;
    __shared__ double red[3 * kW];
    const double2* T = tab + (size_t)blockIdx.x * N;
    double m0 = 0, m1 = 0, t0 = 0;
    bool bad = false;
    for (int i = kBoundFrom + threadIdx.x; i < N; i += kBoundsThreads) {
        const double2 t = T[i];
        const double w = R2 * t.y;
        const double v = fabs(t.x) * w;
        if (!(v == v) || !(w == w)) bad = true;
        m0 = v > m0 ? v : m0;
        m1 = w > m1 ? w : m1;
        if (i >= kTinyFrom) t0 = v > t0 ? v : t0;
    }
    if (bad) { m0 = INFINITY; m1 = INFINITY; t0 = INFINITY; }
    for (int off = 32; off > 0; off >>= 1) {
        const double o0 = __shfl_xor(m0, off), o1 = __shfl_xor(m1, off), o2 = __shfl_xor(t0, off);
        m0 = o0 > m0 ? o0 : m0;
        m1 = o1 > m1 ? o1 : m1;
        t0 = o2 > t0 ? o2 : t0;
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = m0; red[kW + (threadIdx.x >> 6)] = m1; red[2 * kW + (threadIdx.x >> 6)] = t0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = red[0], b = red[kW], c = red[2 * kW];
        for (int w = 1; w < kW; ++w) { a = fmax(a, red[w]); b = fmax(b, red[kW + w]); c = fmax(c, red[2 * kW + w]); }
        bounds[(size_t)blockIdx.x * bstride] = double2{a, b};
        bounds[(size_t)blockIdx.x * bstride + bstride - 1] = double2{c, b};   // series reciprocal: max |veff| R2 e2 over i >= kTinyFrom
    }
}

// per slot and aligned block b of kPipeChunk points (i = b CH + 1 .. (b+1) CH): { min, max } of veff (NaNs are dropped)
__global__ void k_block_minmax(const double2* __restrict__ tab, int N, double2* __restrict__ bounds, int bstride)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= bstride - 2) return;
    const double2* T = tab + (size_t)blockIdx.y * N;
    double mn = INFINITY, mx = -INFINITY;
    for (int i = b * kPipeChunk + 1; i <= (b + 1) * kPipeChunk && i < N; ++i) { mn = fmin(mn, T[i].x); mx = fmax(mx, T[i].x); }
    double2 o;
    o.x = mn;
    o.y = mx;
    bounds[(size_t)blockIdx.y * bstride + 1 + b] = o;
}

// ---- match kernel (Numerov.h:403-504) --------------------------------------------------------------------
// One workgroup of two waves per trial.  Lane 0 of the first wave integrates inward from the cut-off, lane 1 outward from
// the nucleus, at the same time and BLINDLY: per step only the recurrence (11 instructions) and one LDS write of the new
// value -- a lone wave issues one instruction of any kind per ~4.5 cycles, so everything else is moved off this path.
// The second wave works one batch of kMB = 64 nodes ahead and one behind: it computes f_i, d_i = 1 - f_i/12 and the refined
// reciprocal of d_i for the next batch of both streams (one node per lane, coalesced table loads), and it drains the
// previous batch: coalesced stores of the 2 x 64 values, and the search for the match point -- the first node where the
// inward solution turns down (the outermost maximum) or blows up (Numerov.h:463-467) -- as one compare per lane and a
// ballot.  The integrator therefore overruns the match point by up to two batches; that is harmless: what it writes
// below the match point is replaced by the outward values in the rescaling loop at the end.  A stream that has finished
// is zeroed (its recurrence then stays 0).
constexpr int kMB = 64;

struct MatchShared {
    v2d fr[2][2][kMB];       // [batch parity][stream][step] = { f, r }
    double dd[2][2][kMB];    // d
    double uo[2][2][kMB];    // values computed by the integrator
    int done0[2], done1[2], quit[2], mp;   // control, double-buffered by batch parity (written by the helper in iteration b, read after its barrier)
};

// The solve for ONE trial by a workgroup of NT threads: wave 0 integrates, wave 1 helps, further waves (the device-side level search of
// persist.inc runs this with the 320 threads of a sweep workgroup) only keep the barriers company.  Returns the match point to every thread.
template <int NT>
__device__ __forceinline__ int match_solve(MatchShared& sh, const double2* __restrict__ T_, const double E, const int steps, const double us0, const double us10,
                                           const double zero1, const GridScalars& gs, const double2* __restrict__ bounds_slot_ /* the slot's bounds, or null */,
                                           double* __restrict__ P_, double* __restrict__ Qt_)
{
    // inside the device-side level search (persist.inc) these arrive as arguments of a real function: no address space, flat accesses --
    // the helper wave's loads ahead of the integrator would then hide nothing (common.h)
    const double2* __restrict__ T = dfta_as_global(T_);
    const double2* __restrict__ bounds_slot = dfta_as_global(bounds_slot_);      // null stays null
    double* __restrict__ P = dfta_as_global(P_);
    double* __restrict__ Qt = dfta_as_global(Qt_);
    const int lane = threadIdx.x & 63;
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // 0: integrator, 1: helper, 2..: idle
    const int N = gs.N;
    const double R2 = 2. * gs.Rp2delta2;
    const double d2p4 = gs.delta2p4;

    // zero beyond the cut-off (Numerov.h:427-428)
    for (int i = steps + 1 + (int)threadIdx.x; i < N; i += NT) P[i] = 0;

    // may the division use the refined reciprocal?  (same range argument as in the sweeps: d in (0.5, 1.5) from the
    // slot bounds, |w| checked every 16 steps with 16 steps of margin)
    bool fast = false;
    if (bounds_slot) {
        const double2 bd = bounds_slot[0];
        fast = (bd.x + fabs(E) * bd.y + d2p4 < 6.0);
    }
    // batch b: inward nodes i0(b) - k, outward nodes j0(b) + k, k = 0 .. 63
    const int i00 = steps - 2, j00 = 2;
    // table rows of batch b (the helper fetches them two batches ahead: a round trip to L2 takes as long as a batch)
    auto fetch = [&](int b, double2& ti, double2& tj) {
        const int ii = max(i00 - b * kMB - lane, 1), jj = min(j00 + b * kMB + lane, N - 1);
        ti = T[ii];
        tj = T[jj];
    };
    auto produce = [&](int b, const double2 ti, const double2 tj) {
        v2d oi, oj;
        oi.x = (ti.x - E) * R2 * ti.y + d2p4;                                // Numerov.h:100
        oj.x = (tj.x - E) * R2 * tj.y + d2p4;
        const double di = 1. - kH2p12 * oi.x, dj = 1. - kH2p12 * oj.x;
        oi.y = div_in_range(1.0, di);                                         // == the refined reciprocal (w = 1: q = r, rem ~ 0)
        oj.y = div_in_range(1.0, dj);
        sh.fr[b & 1][0][lane] = oi;
        sh.fr[b & 1][1][lane] = oj;
        sh.dd[b & 1][0][lane] = di;
        sh.dd[b & 1][1][lane] = dj;
    };
    if (threadIdx.x == 0) { sh.done0[1] = (i00 < 1) ? 1 : 0; sh.done1[1] = 0; sh.quit[1] = 0; sh.mp = 2; }   // "iteration -1"; matchPoint default (Numerov.h:449)
    double2 nti = {0., 0.}, ntj = {0., 0.};      // helper: rows of the batch after the next one
    if (role == 1) {
        double2 ti, tj;
        fetch(0, ti, tj);
        fetch(1, nti, ntj);
        produce(0, ti, tj);
    }
    __syncthreads();

    if (role == 0) {
        // ---------------- integrator
        double w = 0, wprev = 0, u = 0, fprev = 0;
        if (lane == 0) {
            const double2 ts = T[steps];
            const double2 t1 = T[steps - 1];
            const double us = us0;
            u = us10;
            P[steps] = us;
            P[steps - 1] = u;
            fprev = f_of(ts.x, ts.y, E, gs);
            wprev = (1 - kH2p12 * fprev) * us;
            fprev = f_of(t1.x, t1.y, E, gs);
            w = (1 - kH2p12 * fprev) * u;
        } else if (lane == 1) {
            const double2 t1 = T[1];
            u = zero1;           // Numerov.h:475
            Qt[0] = 0;
            Qt[1] = u;
            fprev = f_of(t1.x, t1.y, E, gs);
            wprev = 0;
            w = (1 - kH2p12 * fprev) * u;
        }
        const int stream = lane == 1 ? 1 : 0;
        for (int b = 0;; ++b) {
            if (lane < 2) {
                const bool mydone = (stream ? sh.done1[(b + 1) & 1] : sh.done0[(b + 1) & 1]) != 0;      // as of iteration b - 1
                if (mydone) { w = 0; wprev = 0; u = 0; fprev = 0; }
                const v2d* __restrict__ mine = &sh.fr[b & 1][stream][0];
                const double* __restrict__ mined = &sh.dd[b & 1][stream][0];
                double* __restrict__ outp = &sh.uo[b & 1][stream][0];
                // first node of this lane's stream in the batch (for the range of the division bounds only)
                const int node0 = stream ? j00 + b * kMB : i00 - b * kMB;
                // groups of 16 steps; the LDS reads of the next group are issued before the chain of the current one starts
                // (reads and writes of this wave queue up in the LDS pipeline: ~4 ns per 16-byte read, ~18 ns per 16-byte
                // write instruction -- as long as a batch's arithmetic unless they overlap with it)
                v2d inA[16], inB[16];
                double dA[16], dB[16];
                auto load16 = [&](v2d (&in)[16], double (&dv)[16], int k0) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) { in[q] = mine[k0 + q]; dv[q] = mined[k0 + q]; }
                };
                auto group = [&](const v2d (&in)[16], const double (&dv)[16], int k0) {
                    // leave the reciprocal path for good when a value gets within 16 steps of the range edges
                    if (fast) {
                        const double au = fabs(u);
                        const bool ok = (au < 1e200 && (au > 1e-250 || u == 0.0));
                        fast = (__ballot(ok) == 3ull);
                    }
                    // the 16 nodes of this group must lie inside the range of the division bounds for the reciprocal path
                    const int lowest = stream ? node0 + k0 : node0 - k0 - 15;
                    const bool grp_fast = fast && (__ballot(mydone || lowest >= kBoundFrom) == 3ull);
                    auto step = [&](const v2d in1, const double d, const int q, const bool use_r) {
                        const double wnext = __builtin_fma(2., w, -wprev) + u * fprev;   // Numerov.h:311 (h2 == 1); 2w is exact
                        wprev = w;
                        w = wnext;
                        if (use_r) {
                            const double qq = wnext * in1.y;
                            const double rem = __builtin_fma(-d, qq, wnext);
                            u = __builtin_fma(rem, in1.y, qq);
                        } else {
                            u = wnext / d;                                              // getU, Numerov.h:510-513
                        }
                        fprev = in1.x;
                        outp[k0 + q] = u;
                    };
                    if (grp_fast) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) step(in[q], dv[q], q, true);
                    } else {
#pragma unroll
                        for (int q = 0; q < 16; ++q) step(in[q], dv[q], q, false);
                    }
                };
                load16(inA, dA, 0);
                load16(inB, dB, 16);
                __builtin_amdgcn_sched_barrier(0);
                group(inA, dA, 0);
                __builtin_amdgcn_sched_barrier(0);
                load16(inA, dA, 32);
                __builtin_amdgcn_sched_barrier(0);
                group(inB, dB, 16);
                __builtin_amdgcn_sched_barrier(0);
                load16(inB, dB, 48);
                __builtin_amdgcn_sched_barrier(0);
                group(inA, dA, 32);
                __builtin_amdgcn_sched_barrier(0);
                group(inB, dB, 48);
            }
            PIPE_BARRIER();      // LDS only: the helper's stores to Psi / the scratch stay in flight
            if (sh.quit[b & 1]) break;
        }
    } else if (role >= 2) {
        // ---------------- not needed: the same barriers as the two working waves
        for (int b = 0;; ++b) {
            PIPE_BARRIER();
            if (sh.quit[b & 1]) break;
        }
    } else {
        // ---------------- helper: drains batch b - 1, produces batch b + 1 while the integrator works on batch b
        double ulast = us10;            // Psi at the node above the first one of the batch being drained
        int found = (i00 < 1) ? 1 : 0, mp = 2;
        bool d1 = false;
        for (int b = 0;; ++b) {
            const double2 cti = nti, ctj = ntj;      // rows of batch b + 1
            fetch(b + 2, nti, ntj);
            if (b >= 1) {
                const int bb = b - 1;
                const int i = i00 - bb * kMB - lane, j = j00 + bb * kMB + lane;
                const double ui = sh.uo[bb & 1][0][lane], uj = sh.uo[bb & 1][1][lane];
                if (!found) {
                    // lane 0: the outermost maximum or a blow-up ends the inward sweep (Numerov.h:463-467)
                    double above = __shfl_up(ui, 1);
                    if (lane == 0) above = ulast;
                    const unsigned long long hits = __ballot(i >= 1 && ((ui < above) || (fabs(ui) > 1E15)));
                    const int first = hits ? __ffsll((long long)hits) - 1 : kMB;
                    if (i >= 1 && lane <= first) P[i] = ui;            // Psi down to the match point
                    if (hits) { found = 1; mp = i00 - bb * kMB - first; }
                    else if (i00 - b * kMB < 1) found = 1;              // ran down to node 1 without a hit: matchPoint stays 2
                    ulast = __shfl(ui, 63);
                }
                if (!d1 && j <= steps) Qt[j] = uj;                     // outward scratch
                // the outward stream is needed up to the match point; while that is unknown, up to the node above the
                // inward stream's position (the match point can only lie below it)
                const int jmax = min(steps, found ? mp : i00 - b * kMB + 1);
                d1 = d1 || (j00 + b * kMB > jmax);
            }
            produce(b + 1, cti, ctj);
            if (lane == 0) {
                sh.done0[b & 1] = found;
                sh.done1[b & 1] = d1 ? 1 : 0;
                sh.mp = mp;
                sh.quit[b & 1] = (found && d1) ? 1 : 0;
            }
            PIPE_BARRIER();      // LDS only: the helper's stores to Psi / the scratch stay in flight
            if (found && d1) break;
        }
    }
    int mp = sh.mp;
    const int lane128 = threadIdx.x;
    __syncthreads();

    // Numerov.h:492-501: value of the outward solution at the match point, rescale the outer part
    double sol;
    if (mp >= 2) sol = Qt[mp];
    else {
        // matchPoint == 1: the reference steps w once from (Psi[1], wprev = 0) and divides by 1 - f(1)/12
        const double2 t1 = T[1];
        const double f1 = f_of(t1.x, t1.y, E, gs);
        const double w1 = (1 - kH2p12 * f1) * zero1;
        const double w2 = 2. * w1 - 0. + zero1 * f1;
        sol = w2 / (1. - kH2p12 * f1);
    }
    // Psi[matchPoint] is the inward value, except for matchPoint == 1 where the outward start value has
    // already overwritten Psi[1] (Numerov.h:475) before the division at Numerov.h:497
    const double factor = sol / (mp >= 2 ? P[mp] : zero1);
    __syncthreads();
    for (int k = lane128; k <= steps; k += NT) {
        double v;
        if (k < mp) v = (k == 0) ? 0.0 : Qt[k];
        else if (k == mp) v = sol;
        else v = P[k] * factor;
        P[k] = v;
    }
    return mp;
}

__global__ __launch_bounds__(128) void k_match(const double2* __restrict__ tab, const int* __restrict__ trial_slot,
                                               const double* __restrict__ Earr, const int* __restrict__ startArr,
                                               const double* __restrict__ usArr, const double* __restrict__ us1Arr,
                                               const int* __restrict__ larr, double zero_l0, double zero_l1, double zero_l2,
                                               double zero_l3, GridScalars gs, const double2* __restrict__ bounds, int bstride,
                                               double* __restrict__ Psi, double* __restrict__ Q, int* __restrict__ matchPoint)
{
    __shared__ MatchShared sh;
    const int t = blockIdx.x;
    const int steps = startArr[t];
    if (steps < 0) return;               // frozen job (levels.hip): Psi of its last solve stands
    const int slot = trial_slot[t];
    const int l = larr[t];
    const double zero1 = l == 0 ? zero_l0 : (l == 1 ? zero_l1 : (l == 2 ? zero_l2 : zero_l3));
    const int mp = match_solve<128>(sh, tab + (size_t)slot * gs.N, Earr[t], steps, usArr[t], us1Arr[t], zero1, gs,
                                    bounds ? bounds + (size_t)slot * bstride : nullptr, Psi + (size_t)t * gs.N, Q + (size_t)t * gs.N);
    if (threadIdx.x == 0) matchPoint[t] = mp;
}

// ---- uniform grid (Numerov.h:16-70 and the IsUniform() branches of Numerov.h:272-504) -------------------------------------
// The reference's uniform-grid path is not reachable from its front end (DFTAtomFrame.cpp:191,194) and not part of any
// BASELINE configuration; it is served by plain kernels: one lane per trial, the wave marches the grid index together, table
// rows staged through LDS 64 at a time with the next 64 in flight.  Arithmetic in the reference's order:
//     f_i = 2 (V_i + l(l+1)/(r_i r_i)/2 - E),   w_i = 2 w_{i+1} - w_{i+2} + h^2 u_{i+1} f_{i+1},   u_i = w_i / (1 - h^2/12 f_i)
template <int KIND>
__device__ __forceinline__ void usweep_wave(const SweepArgs& a, const GridScalars& gs, const int b, const int lane, double2* rows)
{
    const int cnt = a.blk_cnt[b];
    const int t = a.blk_first[b] + (lane < cnt ? lane : 0);
    const bool valid = (lane < cnt) && (a.start[t] >= 2);
    if (__ballot(valid) == 0ull) return;
    const int slot = a.blk_slot[b];
    const double2* __restrict__ T = a.tab + (size_t)slot * gs.N;
    const unsigned l = static_cast<unsigned>(a.slot_l[slot]);
    const double E = a.E[t];
    const int start = valid ? a.start[t] : 2;
    const int limit = (KIND == DFTA_SWEEP_COUNT) ? a.limit[t] : 0;
    const double h = gs.h, h2 = gs.h2, h2p12 = gs.h2p12;
    // prologue (Numerov.h:289-301): the first two points sit at startPoint and startPoint - h, not at i h
    double u, w, wprev, fprev, prevSol;
    {
        const double sp = uniform_start_point(E, gs.Rmax);
        const double p1 = sp - h;
        const double us = a.us[t];
        u = a.us1[t];
        const double fs = 2. * ((T[start].y + l * (l + 1.) / (sp * sp) * 0.5) - E);           // Numerov.h:21-30
        wprev = (1 - h2p12 * fs) * us;
        fprev = 2. * ((T[start - 1].y + l * (l + 1.) / (p1 * p1) * 0.5) - E);
        w = (1 - h2p12 * fprev) * u;
        prevSol = us;
    }
    bool oldSgn = (u > 0), live = valid, exited = false, flag = false;
    int count = 0, trips = 0;
    const int my_hi = valid ? start - 2 : 0;
    int ihi = my_hi;
    for (int off = 32; off > 0; off >>= 1) ihi = max(ihi, __shfl_xor(ihi, off));
    ihi = __builtin_amdgcn_readfirstlane(ihi);
    double2 nxt = T[max(ihi - lane, 1)];
    for (int top = ihi; top >= 1; top -= 64) {
        __syncthreads();
        rows[lane] = nxt;
        nxt = T[max(top - 64 - lane, 1)];
        __syncthreads();
        const int nk = top < 64 ? top : 64;
        for (int k = 0; k < nk; ++k) {
            const int i = top - k;
            const double2 tv = rows[k];
            if (live && i <= my_hi) {
                const double wnext = 2. * w - wprev + h2 * u * fprev;          // Numerov.h:311
                wprev = w;
                w = wnext;
                const double f = 2. * (tv.x - E);                               // Numerov.h:28-30 at position = h i
                prevSol = u;
                u = w / (1. - h2p12 * f);                                       // getU, Numerov.h:510-513
                fprev = f;
                ++trips;
                if (KIND == DFTA_SWEEP_COUNT) {                                 // Numerov.h:323-340
                    if (fabs(u) == INFINITY) { live = false; exited = true; }
                    else {
                        const bool newSgn = (u > 0);
                        if (newSgn != oldSgn) {
                            ++count;
                            if (count > limit) { live = false; exited = true; }
                            oldSgn = newSgn;
                        }
                        if (live) {
                            if (tv.x <= E) flag = true;
                            else if (flag && tv.x > E) { live = false; exited = true; }
                        }
                    }
                }
            }
        }
        if (KIND == DFTA_SWEEP_COUNT && __ballot(live) == 0ull) break;
    }
    double u0 = NAN;
    if (valid && (KIND == DFTA_SWEEP_ZERO || !exited)) {
        u0 = u * (2 + h2 * fprev) - prevSol;                                    // Numerov.h:345 / 398
        if (KIND == DFTA_SWEEP_COUNT && (u0 > 0) != oldSgn) ++count;            // Numerov.h:346-347
    }
    if (valid) {
        if (KIND == DFTA_SWEEP_COUNT) a.count[t] = count;
        if (a.u0) a.u0[t] = u0;
        if (a.trip) a.trip[t] = trips;
        if (KIND == DFTA_SWEEP_COUNT && a.phi) a.phi[t] = NAN;                   // no secant samples on this path
        if (KIND == DFTA_SWEEP_COUNT && a.istop) a.istop[t] = -1;
    }
    if (a.total_trips) {
        int sum = valid ? trips : 0;
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        if (lane == 0) atomicAdd(a.total_trips, (unsigned long long)sum);
    }
}

__global__ __launch_bounds__(64) void k_usweep(SweepArgs a, GridScalars gs, int nwaves)
{
    __shared__ double2 rows[64];
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    if (b >= nwaves) return;
    const int kind = a.blk_kind ? __builtin_amdgcn_readfirstlane(a.blk_kind[b]) : a.kind;
    if (kind == DFTA_SWEEP_COUNT) usweep_wave<DFTA_SWEEP_COUNT>(a, gs, b, lane, rows);
    else                          usweep_wave<DFTA_SWEEP_ZERO>(a, gs, b, lane, rows);
}

// SolveSchrodingerMatchSolutionCompletely on the uniform grid (Numerov.h:403-504 with IsUniform()): one 64-thread block
// per trial; thread 0 integrates (inward to the match point, then outward), the block stages table rows and stores Psi.
// The step is re-derived from the truncated step count (Numerov.h:430), so positions are h' i with a per-trial h'.
__global__ __launch_bounds__(64) void k_umatch(const double2* __restrict__ tab, const int* __restrict__ trial_slot,
                                               const double* __restrict__ Earr, const int* __restrict__ startArr,
                                               const double* __restrict__ usArr, const double* __restrict__ us1Arr,
                                               const double* __restrict__ uzArr, const int* __restrict__ larr, GridScalars gs,
                                               double* __restrict__ Psi, int* __restrict__ matchPoint)
{
    __shared__ double2 rows[64];
    __shared__ double outv[64];
    __shared__ int s_n, s_stop, s_mp;
    __shared__ double s_factor;
    const int t = blockIdx.x, tid = threadIdx.x;
    const int steps = startArr[t];
    if (steps < 0) return;                      // frozen job (levels.hip)
    const int N = gs.N;
    const double2* __restrict__ T = tab + (size_t)trial_slot[t] * N;
    double* __restrict__ P = Psi + (size_t)t * N;
    const double E = Earr[t];
    const unsigned l = static_cast<unsigned>(larr[t]);
    for (int i = steps + 1 + tid; i < N; i += 64) P[i] = 0;                     // Numerov.h:427-428
    const double sp = uniform_start_point(E, gs.Rmax);
    const double h = sp / steps, h2 = h * h, h2p12 = h2 / 12.;                   // Numerov.h:430-432
    const double ll = l * (l + 1.);
    auto func = [&](double V, double position) { return 2. * ((V + ll / (position * position) * 0.5) - E); };   // Numerov.h:21-30
    double sol = 0, w = 0, wprev = 0, f = 0, above = 0;
    if (tid == 0) {
        sol = usArr[t];
        P[steps] = sol;
        f = func(T[steps].y, sp);
        wprev = (1 - h2p12 * f) * sol;
        sol = us1Arr[t];
        P[steps - 1] = sol;
        f = func(T[steps - 1].y, sp - h);
        w = (1 - h2p12 * f) * sol;
        above = sol;
        s_stop = 0;
        s_mp = 2;                                                                // Numerov.h:449
    }
    __syncthreads();
    for (int top = steps - 2; top >= 1; top -= 64) {                             // inward, Numerov.h:450-470
        const int idx = top - tid;
        if (idx >= 1) rows[tid] = T[idx];
        __syncthreads();
        if (tid == 0) {
            int n = 0;
            for (int k = 0; k < 64 && top - k >= 1; ++k) {
                const int i = top - k;
                const double wnext = 2. * w - wprev + h2 * sol * f;
                wprev = w;
                w = wnext;
                f = func(rows[k].y, h * i);
                sol = w / (1. - h2p12 * f);
                outv[k] = sol;
                n = k + 1;
                if (sol < above || fabs(sol) > 1E15) { s_mp = i; s_stop = 1; break; }
                above = sol;
            }
            s_n = n;
        }
        __syncthreads();
        if (tid < s_n) P[top - tid] = outv[tid];
        const int stop = s_stop;
        __syncthreads();
        if (stop) break;
    }
    const int mp = s_mp;
    __syncthreads();
    if (tid == 0) {                                                              // outward, Numerov.h:472-480
        P[0] = 0;
        wprev = 0;
        sol = uzArr[t];
        P[1] = sol;
        f = func(T[1].y, h);
        w = (1 - h2p12 * f) * sol;
    }
    for (int base = 2; base < mp; base += 64) {                                  // Numerov.h:482-492
        const int idx = base + tid;
        if (idx < mp) rows[tid] = T[idx];
        __syncthreads();
        if (tid == 0) {
            int n = 0;
            for (int k = 0; k < 64 && base + k < mp; ++k) {
                const int i = base + k;
                const double wnext = 2. * w - wprev + h2 * sol * f;
                wprev = w;
                w = wnext;
                f = func(rows[k].y, h * i);
                sol = w / (1. - h2p12 * f);
                outv[k] = sol;
                n = k + 1;
            }
            s_n = n;
        }
        __syncthreads();
        if (tid < s_n) P[base + tid] = outv[tid];
        __syncthreads();
    }
    if (tid == 0) {                                                              // Numerov.h:494-499
        w = 2. * w - wprev + h2 * sol * f;
        f = func(T[mp].y, h * mp);
        sol = w / (1. - h2p12 * f);
        __threadfence_block();
        s_factor = sol / P[mp];
        P[mp] = sol;
        matchPoint[t] = mp;
    }
    __syncthreads();
    const double factor = s_factor;
    for (int i = mp + 1 + tid; i <= steps; i += 64) P[i] *= factor;              // Numerov.h:500-501
}

GridScalars scalars_of(const dfta_grid* g)
{
    GridScalars gs;
    gs.N = g->N; gs.delta = g->delta; gs.Rp2delta2 = g->Rp2delta2; gs.delta2p4 = g->delta2p4; gs.far_thr = g->far_arg_threshold;
    gs.uniform = g->uniform; gs.Rmax = g->Rmax; gs.h = g->h; gs.h2 = g->h2; gs.h2p12 = g->h2p12;
    return gs;
}

// uniform grid, host side: cut-off and start values exactly as the reference evaluates them (libm), Numerov.h:32-41,274-296
void host_boundary_uniform(const dfta_grid* g, double E, unsigned l, bool for_match, int* start, double* us, double* us1, double* uz)
{
    const double s = sqrt(2. * fabs(E));
    const double mr = 200. / s;
    const double sp = mr < g->Rmax ? mr : g->Rmax;
    const long steps = static_cast<long>(sp / g->h);
    const double hh = for_match ? sp / steps : g->h;
    *start = static_cast<int>(steps);
    *us = exp(-sp * s);
    *us1 = exp(-(sp - hh) * s);
    if (uz) *uz = pow(hh, static_cast<double>(l) + 1.);
}

// host-side boundary values exactly as the reference evaluates them (libm exp)
void host_boundary(const dfta_grid* g, double E, int* start, double* us, double* us1)
{
    const double s = sqrt(2. * fabs(E));
    auto far = [&](int i) { return exp(-g->h_r[i] * s - static_cast<double>(i) * g->delta * 0.5); };
    size_t maxIndex = static_cast<size_t>(g->N - 1), minIndex = 1;
    while (maxIndex - minIndex > 1) {
        const size_t mid = (maxIndex + minIndex) / 2;
        if (far(static_cast<int>(mid)) < 1E-200) maxIndex = mid; else minIndex = mid;
    }
    *start = static_cast<int>(maxIndex);
    *us = far(static_cast<int>(maxIndex));
    *us1 = far(static_cast<int>(maxIndex) - 1);
}


#include "levels_device.inc"
#include "persist.inc"
#include "own.inc"

}  // namespace

// ---- own-pace level search of a batch (own.inc): one workgroup of W waves per live level, ONE ordinary launch ---------------------------
// d_live (device, may be null: level q = job q): the jobs to solve; their records carry phase = first bisection and tbase = q * 64 W.
// The trial arrays are the level solver's own (room for nlive * 64 W trials; blk_first[b] = 64 b, blk_cnt[b] = 64).
int dfta_launch_levels_own(dfta_ctx* ctx, const dfta_grid* g, dfta::Job* d_jobs, const int* d_live, int nlive, int W, const double2* d_tab, const double2* d_bounds,
                           int* blk_slot, const int* blk_first, const int* blk_cnt, double* dE, int* dLimit, int* dStart, double* dUs, double* dUs1, int* dCount,
                           double* dU0, double* dPhi, int* dIstop, int* dTrip, unsigned long long* d_counters, bool stats, int nopredict, int spine_cap)
{
    if (nlive < 1 || W < 1 || W > 8 || (W & (W - 1)) != 0 || g->uniform) return DFTA_ERR_INVALID;
    SweepArgs a;
    a.slot_l = nullptr;
    a.phi = dPhi; a.istop = dIstop;
    a.kind = DFTA_SWEEP_COUNT; a.blk_kind = nullptr; a.bounds = d_bounds; a.bstride = dfta_bounds_stride(g);
    a.tab = d_tab; a.blk_slot = blk_slot; a.blk_first = blk_first; a.blk_cnt = blk_cnt;
    a.E = dE; a.limit = dLimit; a.start = dStart; a.us = dUs; a.us1 = dUs1; a.count = dCount; a.u0 = dU0;
    a.trip = stats ? dTrip : nullptr; a.total_trips = stats ? d_counters + 1 : nullptr;
    OwnArgs oa;
    oa.jobs = d_jobs; oa.live = d_live; oa.r = g->d_r; oa.W = W; oa.nopredict = nopredict; oa.spine_cap = spine_cap;
    oa.E = dE; oa.us = dUs; oa.us1 = dUs1; oa.limit = dLimit; oa.start = dStart; oa.blk_slot = blk_slot;
    oa.issued = d_counters;
    oa.max_rounds = reinterpret_cast<unsigned int*>(d_counters + 2);
    hipLaunchKernelGGL(k_levels_own, dim3(nlive), dim3(64 * W), 0, ctx->stream, a, scalars_of(g), oa);
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

// ---- device-side level search (persist.inc): buffers and launch ------------------------------------------------------------------
void dfta_persist_destroy(dfta_persist_buffers* pb)
{
    if (!pb) return;
    for (void* q : {(void*)pb->d_ctl, (void*)pb->E, (void*)pb->us, (void*)pb->us1, (void*)pb->u0, (void*)pb->phi, (void*)pb->limit, (void*)pb->start,
                    (void*)pb->count, (void*)pb->istop, (void*)pb->trip, (void*)pb->blk, (void*)pb->candP, (void*)pb->candQ}) if (q) (void)hipFree(q);
    *pb = dfta_persist_buffers();
}

int dfta_persist_create(dfta_ctx* ctx, const dfta_grid* g, int nlive_cap, dfta_persist_buffers* pb)
{
    dfta_persist_destroy(pb);
    int nblocks = std::min(ctx->num_cu, kPersistMaxBlocks);
    if (const char* e = dfta_knob("LEVELS_PERSIST_BLOCKS")) nblocks = std::max(2, std::min(atoi(e), nblocks));     // measurements
    pb->nblocks = nblocks;
    pb->tmax = 64 * nblocks;                       // a level never has more trials in a round than the machine has lanes
    pb->nlive_cap = std::min(nlive_cap, kPersistMaxJobs);
    pb->trace_cap = 4096;
    pb->fault_block = dfta_knob("FAULT_PERSIST_WORKER") ? 1 : -1;
    pb->timeout_ms = dfta_knob("LEVELS_PERSIST_TIMEOUT_MS") ? atof(dfta_knob("LEVELS_PERSIST_TIMEOUT_MS")) : 0.0;
    pb->plain_launch = dfta_knob("LEVELS_PERSIST_PLAIN_LAUNCH") != nullptr;
    pb->equal_shares = dfta_knob("LEVELS_PERSIST_EQUAL") != nullptr;
    pb->want_trace = dfta_knob("LEVELS_PERSIST_TRACE") != nullptr;
    const size_t nt = (size_t)pb->nlive_cap * pb->tmax;
    pb->ctl_bytes = sizeof(PersistCtl) + sizeof(unsigned long long) * kPersistMaxBlocks + sizeof(PersistJob) * kPersistMaxJobs;
    hipError_t e = hipMalloc(&pb->d_ctl, pb->ctl_bytes + sizeof(unsigned long long) * 4 * pb->trace_cap);
    auto alloc = [&](auto& ptr, size_t count) { if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ptr), sizeof(*ptr) * count); };
    alloc(pb->E, nt); alloc(pb->us, nt); alloc(pb->us1, nt); alloc(pb->u0, nt); alloc(pb->phi, nt);
    alloc(pb->limit, nt); alloc(pb->start, nt); alloc(pb->count, nt); alloc(pb->istop, nt); alloc(pb->trip, nt);
    alloc(pb->blk, (size_t)3 * kPersistMaxBlocks);
    // speculative matches of the last round's candidate eigenvalues: a wavefunction and a scratch vector per workgroup (0.5 GB at 131 073 nodes, 4.3 GB at 1 048 577)
    if ((size_t)nblocks * g->N * 16 <= ((size_t)8 << 30) && dfta_knob("LEVELS_PERSIST_NOCAND") == nullptr) { alloc(pb->candP, (size_t)nblocks * g->N); alloc(pb->candQ, (size_t)nblocks * g->N); }
    if (e == hipSuccess) e = hipMemset(pb->count, 0, sizeof(int) * nt);
    if (e == hipSuccess) e = hipMemset(pb->u0, 0, sizeof(double) * nt);
    if (e == hipSuccess) e = hipMemset(pb->trip, 0, sizeof(int) * nt);
    if (e != hipSuccess) {
        snprintf(ctx->err, sizeof(ctx->err), "device-side level search: %s", hipGetErrorString(e));
        dfta_persist_destroy(pb);
        return DFTA_ERR_HIP;
    }
    return DFTA_OK;
}

// Launches the search of the live levels `live` (host, indices into d_jobs, whose records carry phase = first bisection, tbase = position
// in `live` x pb->tmax).  *aborted = 1: a worker was lost (time-out) -- nothing is valid, the caller repeats the solve with host rounds.
int dfta_launch_levels_persist(dfta_ctx* ctx, const dfta_grid* g, dfta_persist_buffers* pb, dfta::Job* d_jobs, const int* live, int nlive,
                               const double2* d_tab, const double2* d_bounds, double* d_Psi, double* d_Q, int* d_jstart_keep,
                               unsigned long long* d_counters, bool stats, int nopredict, int integ_rule, const double* tuning /* noise rel, abs, secant, kappa */,
                               int fixed_point, int* rounds, int* aborted, std::vector<unsigned long long>* trace_out, const int* share, int deep_reserve)
{
    *aborted = 0;
    if (nlive < 1 || nlive > pb->nlive_cap || g->uniform) return DFTA_ERR_INVALID;
    const int nblocks = pb->nblocks;
    const int base = nblocks / nlive;
    if (base < 1) return DFTA_ERR_INVALID;
    hipStream_t st = ctx->stream;
    {   // this translation unit's copies of the prediction constants (levels_device.inc), per device
        static std::mutex mu;                 // (contexts of several host threads may launch on the same device)
        std::lock_guard<std::mutex> lock(mu);
        static double last[16][4];
        static int last_fp[16];
        static bool have[16];
        const int dv = ctx->device >= 0 && ctx->device < 16 ? ctx->device : -1;
        if (dv < 0 || !have[dv] || memcmp(last[dv], tuning, sizeof(last[dv])) != 0 || last_fp[dv] != fixed_point) {
            DFTA_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_noise_rel), &tuning[0], sizeof(double)));
            DFTA_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_noise_abs), &tuning[1], sizeof(double)));
            DFTA_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_secant_noise), &tuning[2], sizeof(double)));
            DFTA_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_secant_kappa), &tuning[3], sizeof(double)));
            DFTA_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_fixed_point), &fixed_point, sizeof(int)));
            if (dv >= 0) { memcpy(last[dv], tuning, sizeof(last[dv])); last_fp[dv] = fixed_point; have[dv] = true; }
        }
    }
    // control block: pool, counters, mailboxes (the first workgroup of every level plans its first round), the levels' workgroups
    std::vector<unsigned char>& hb = pb->h_stage;
    const size_t ctl_used = sizeof(PersistCtl) + sizeof(unsigned long long) * kPersistMaxBlocks + sizeof(PersistJob) * (size_t)nlive;   // (the records of the levels in use)
    hb.assign(ctl_used, 0);
    PersistCtl* hc = reinterpret_cast<PersistCtl*>(hb.data());
    unsigned long long* hm = reinterpret_cast<unsigned long long*>(hb.data() + sizeof(PersistCtl));
    PersistJob* hj = reinterpret_cast<PersistJob*>(hb.data() + sizeof(PersistCtl) + sizeof(unsigned long long) * kPersistMaxBlocks);
    hc->live = (unsigned)nlive;
    hc->t0 = ~0ull;
    int next = 0;
    for (int k = 0; k < nlive; ++k) {
        // (a level never starts with more than an equal share -- except where that share is ONE workgroup: the caller hands the rest out as second ones)
        const int mine = share ? (base == 1 ? std::min(std::max(share[k], 1), 2) : std::max(2, std::min(share[k], base))) : base;      // (a level never starts with more than an equal share)
        hj[k].job = live[k];
        hj[k].base = mine;
        hj[k].nown = mine;
        for (int q = 0; q < mine; ++q) hj[k].blocks[q] = static_cast<unsigned short>(next + q);
        hm[next] = persist_msg(kCmdPlan, k, 0);
        next += mine;
    }
    if (next > nblocks) return DFTA_ERR_INVALID;
    for (int q = next; q < nblocks; ++q) hc->pool[q >> 6] |= 1ull << (q & 63);
    unsigned char* dctl = static_cast<unsigned char*>(pb->d_ctl);
    DFTA_HIP(ctx, hipMemcpyAsync(dctl, hb.data(), ctl_used, hipMemcpyHostToDevice, st));

    SweepArgs a;
    a.slot_l = nullptr;
    a.phi = pb->phi; a.istop = pb->istop;
    a.kind = DFTA_SWEEP_COUNT; a.blk_kind = nullptr; a.bounds = d_bounds; a.bstride = dfta_bounds_stride(g);
    a.tab = d_tab; a.blk_slot = pb->blk; a.blk_first = pb->blk + kPersistMaxBlocks; a.blk_cnt = pb->blk + 2 * kPersistMaxBlocks;
    a.E = pb->E; a.limit = pb->limit; a.start = pb->start; a.us = pb->us; a.us1 = pb->us1; a.count = pb->count; a.u0 = pb->u0;
    a.trip = stats ? pb->trip : nullptr; a.total_trips = stats ? d_counters + 1 : nullptr;
    PersistArgs pa;
    pa.jobs = d_jobs;
    pa.ctl = reinterpret_cast<PersistCtl*>(dctl);
    pa.mbox = reinterpret_cast<unsigned long long*>(dctl + sizeof(PersistCtl));
    pa.pj = reinterpret_cast<PersistJob*>(dctl + sizeof(PersistCtl) + sizeof(unsigned long long) * kPersistMaxBlocks);
    pa.r = g->d_r;
    pa.nblocks = nblocks;
    pa.nopredict = nopredict;
    pa.timeout_ticks = static_cast<long long>(100e6 * (3.0 + 4.0 * g->N / 131072.0));      // wall clock at 100 MHz
    if (pb->timeout_ms > 0) pa.timeout_ticks = static_cast<long long>(1e5 * pb->timeout_ms);
    pa.E = pb->E; pa.us = pb->us; pa.us1 = pb->us1; pa.limit = pb->limit; pa.start = pb->start;
    pa.blk_slot = pb->blk; pa.blk_first = pb->blk + kPersistMaxBlocks; pa.blk_cnt = pb->blk + 2 * kPersistMaxBlocks;
    pa.Psi = d_Psi; pa.Q = d_Q; pa.jstart_keep = d_jstart_keep;
    pa.candP = pb->candP; pa.candQ = pb->candQ;
    pa.eh = g->d_eh; pa.cnst = g->d_cnst;
    for (int q = 0; q < 4; ++q) pa.zero1[q] = g->zero1[q];
    pa.step = 1.0;
    pa.rule = integ_rule;
    pa.issued = d_counters;
    pa.trace = trace_out ? reinterpret_cast<unsigned long long*>(dctl + pb->ctl_bytes) : nullptr;
    pa.trace_cap = (unsigned)pb->trace_cap;
    pa.fault_block = pb->fault_block;
    pa.deep_reserve = deep_reserve;
    GridScalars gs = scalars_of(g);
    // the workers wait for each other: co-residency is the launch's business (one workgroup per compute unit: 140 KB of LDS).  Under a
    // profiler the launch is an ordinary one (rocprofiler-sdk 7.2 crashes in an exit handler after a cooperative launch, see poisson.hip)
    const bool plain = getenv("ROCP_TOOL_LIBRARIES") != nullptr || pb->plain_launch;
    if (plain) {
        hipLaunchKernelGGL(k_levels_persist, dim3(nblocks), dim3(kPipeThreads), 0, st, a, gs, pa);
        DFTA_CHECK_LAUNCH(ctx);
    } else {
        void* args[] = {&a, &gs, &pa};
        const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_levels_persist), dim3(nblocks), dim3(kPipeThreads), args, 0, st);
        if (e != hipSuccess) { (void)hipGetLastError(); *aborted = 1; return DFTA_OK; }      // not co-resident: host rounds
    }
    PersistCtl out;
    DFTA_HIP(ctx, hipMemcpyAsync(&out, dctl, sizeof(out), hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    if (out.abort || out.live != 0) { *aborted = 1; return DFTA_OK; }
    if (rounds) *rounds = (int)out.max_rounds;
    if (trace_out) {
        const unsigned int n = std::min(out.trace_n, (unsigned)pb->trace_cap);
        trace_out->resize((size_t)4 * n);
        if (n) DFTA_HIP(ctx, hipMemcpy(trace_out->data(), dctl + pb->ctl_bytes, sizeof(unsigned long long) * 4 * n, hipMemcpyDeviceToHost));
    }
    return DFTA_OK;
}

// per slot: [0] the fast-division bounds, [1 ..] {min, max} of veff per block of kPipeChunk points, [stride-1] the bounds over
// i >= kTinyFrom (series reciprocal)
int dfta_bounds_stride(const dfta_grid* g) { return 1 + (g->N + kPipeChunk - 1) / kPipeChunk + 1 + 1; }

// Launch plumbing shared with levels.hip ------------------------------------------------------------------
int dfta_launch_build_tab(dfta_ctx* ctx, const dfta_grid* g, double2* tab, const double* dV, const int* d_slot_v,
                          const int* d_slot_l, int nslots, double2* bounds)
{
    dim3 grid((g->N + 255) / 256 > 64 ? 64 : (g->N + 255) / 256, nslots);
    hipLaunchKernelGGL(k_build_tab, grid, dim3(256), 0, ctx->stream, tab, dV, g->d_cl, g->d_e2, d_slot_v, d_slot_l, g->N, g->uniform);
    DFTA_CHECK_LAUNCH(ctx);
    if (bounds && !g->uniform) {
        const int bstride = dfta_bounds_stride(g);
        hipLaunchKernelGGL(k_slot_bounds, dim3(nslots), dim3(kBoundsThreads), 0, ctx->stream, tab, g->N, 2. * g->Rp2delta2, bounds, bstride);
        hipLaunchKernelGGL(k_block_minmax, dim3((bstride + 254) / 256, nslots), dim3(256), 0, ctx->stream, tab, g->N, bounds, bstride);
        DFTA_CHECK_LAUNCH(ctx);
    }
    return DFTA_OK;
}

int dfta_launch_boundary(dfta_ctx* ctx, const dfta_grid* g, const double* dE, int ntrials, int* dStart, double* dUs, double* dUs1,
                         int for_match, const int* dL, double* dUz, hipStream_t stream)
{
    hipStream_t st = stream ? stream : ctx->stream;
    if (g->uniform) {
        hipLaunchKernelGGL(k_boundary_uniform, dim3((ntrials + 255) / 256), dim3(256), 0, st, dE, ntrials, scalars_of(g), dStart,
                           dUs, dUs1, for_match, dL, dL ? dUz : nullptr);
        DFTA_CHECK_LAUNCH(ctx);
        return DFTA_OK;
    }
    hipLaunchKernelGGL(k_boundary, dim3((ntrials + 255) / 256), dim3(256), 0, st, g->d_r, dE, ntrials,
                       scalars_of(g), dStart, dUs, dUs1);
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

int dfta_launch_sweep(dfta_ctx* ctx, const dfta_grid* g, int kind, const int* blk_kind, int nblocks, const double2* tab,
                      const int* blk_slot, const int* blk_first, const int* blk_cnt, const double* dE, const int* dLimit,
                      const int* dStart, const double* dUs, const double* dUs1, int* dCount, double* dU0, int* dTrip,
                      unsigned long long* dTotalTrips, const double2* bounds, double* dPhi, int* dIstop, const int* d_slot_l, const int* d_queue, int qcap)
{
    SweepArgs a;
    a.slot_l = d_slot_l;
    a.phi = dPhi; a.istop = dIstop;
    a.kind = kind; a.blk_kind = blk_kind; a.bounds = bounds; a.bstride = dfta_bounds_stride(g);
    a.tab = tab; a.blk_slot = blk_slot; a.blk_first = blk_first; a.blk_cnt = blk_cnt; a.E = dE; a.limit = dLimit;
    a.start = dStart; a.us = dUs; a.us1 = dUs1; a.count = dCount; a.u0 = dU0; a.trip = dTrip; a.total_trips = dTotalTrips;
    const bool pipe = ctx->sweep_kernel == DFTA_SWEEP_AUTO ? nblocks <= kPipeMaxBlocks : ctx->sweep_kernel == DFTA_SWEEP_PIPELINED;
    if (g->uniform) {
        if (!d_slot_l) { snprintf(ctx->err, sizeof(ctx->err), "uniform sweeps need the slots' l"); return DFTA_ERR_INVALID; }
        hipLaunchKernelGGL(k_usweep, dim3(nblocks), dim3(64), 0, ctx->stream, a, scalars_of(g), nblocks);
    } else if (pipe) {
        hipLaunchKernelGGL((k_sweep_pipe<kPipeChunk>), dim3(nblocks), dim3(kPipeThreads), 0, ctx->stream, a, scalars_of(g), nblocks, d_queue,
                           d_queue ? d_queue + kSweepQueueClasses + 1 : nullptr, qcap);
    } else if (d_queue) {
        hipLaunchKernelGGL((k_sweep_queue<kChunk>), dim3(nblocks), dim3(64), 0, ctx->stream, a, scalars_of(g), d_queue, d_queue + kSweepQueueClasses + 1, qcap);
    } else {
        const dim3 grid((nblocks + 3) / 4), block(256);
        hipLaunchKernelGGL((k_sweep<kChunk>), grid, block, 0, ctx->stream, a, scalars_of(g), nblocks);
    }
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

bool dfta_sweep_is_fused(const dfta_ctx* ctx, int nblocks)
{
    return !(ctx->sweep_kernel == DFTA_SWEEP_AUTO ? nblocks <= kPipeMaxBlocks : ctx->sweep_kernel == DFTA_SWEEP_PIPELINED);
}

int dfta_launch_match(dfta_ctx* ctx, const dfta_grid* g, int ntrials, const double2* tab, const int* d_trial_slot,
                      const double* dE, const int* dStart, const double* dUs, const double* dUs1, const int* dL,
                      double* dPsi, double* dQ, int* dMatch, const double2* bounds, const double* dUz, hipStream_t stream)
{
    hipStream_t st = stream ? stream : ctx->stream;
    if (g->uniform) {
        if (!dUz) { snprintf(ctx->err, sizeof(ctx->err), "uniform match needs the start values at the origin"); return DFTA_ERR_INVALID; }
        hipLaunchKernelGGL(k_umatch, dim3(ntrials), dim3(64), 0, st, tab, d_trial_slot, dE, dStart, dUs, dUs1, dUz, dL, scalars_of(g),
                           dPsi, dMatch);
        DFTA_CHECK_LAUNCH(ctx);
        return DFTA_OK;
    }
    hipLaunchKernelGGL(k_match, dim3(ntrials), dim3(128), 0, st, tab, d_trial_slot, dE, dStart, dUs, dUs1, dL,
                       g->zero1[0], g->zero1[1], g->zero1[2], g->zero1[3], scalars_of(g), bounds, dfta_bounds_stride(g), dPsi, dQ,
                       dMatch);
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

// ---- C ABI -----------------------------------------------------------------------------------------------------
namespace {

struct Grouping {
    std::vector<int> order;       // sorted trial -> original trial
    std::vector<int> slot_v, slot_l;
    std::vector<int> blk_slot, blk_first, blk_cnt;
    std::vector<int> trial_slot;  // per sorted trial
};

// group trials by (vidx, l) so every wave shares its per-point inputs
int make_grouping(int ntrials, const int* vidx, const int* l, int nV, Grouping& G)
{
    G.order.resize(ntrials);
    std::iota(G.order.begin(), G.order.end(), 0);
    auto key = [&](int t) { return (vidx ? vidx[t] : 0) * 4 + l[t]; };
    for (int t = 0; t < ntrials; ++t) {
        const int v = vidx ? vidx[t] : 0;
        if (v < 0 || v >= nV || l[t] < 0 || l[t] > 3) return DFTA_ERR_INVALID;
    }
    std::stable_sort(G.order.begin(), G.order.end(), [&](int a, int b) { return key(a) < key(b); });
    G.trial_slot.resize(ntrials);
    int p = 0;
    while (p < ntrials) {
        const int k = key(G.order[p]);
        int q = p;
        while (q < ntrials && key(G.order[q]) == k) ++q;
        const int slot = static_cast<int>(G.slot_v.size());
        G.slot_v.push_back(k / 4);
        G.slot_l.push_back(k % 4);
        for (int s = p; s < q; s += 64) {
            G.blk_slot.push_back(slot);
            G.blk_first.push_back(s);
            G.blk_cnt.push_back(std::min(64, q - s));
        }
        for (int s = p; s < q; ++s) G.trial_slot[s] = slot;
        p = q;
    }
    return DFTA_OK;
}

template <typename T>
hipError_t upload(DevBuf<T>& d, const std::vector<T>& h, hipStream_t s)
{
    hipError_t e = d.alloc(h.size());
    if (e != hipSuccess || h.empty()) return e;
    return hipMemcpyAsync(d.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s);
}

}  // namespace

extern "C" int dfta_numerov_sweeps(dfta_ctx* ctx, const dfta_grid* g, int kind, int boundary, int nV, const double* V,
                                   int ntrials, const int* vidx, const int* l, const double* E, const int* nodesLimit,
                                   int* count_out, double* u0_out, int* start_out, int* trip_out)
{
    if (!ctx || !g) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, ntrials >= 0, "negative trial count");
    if (ntrials == 0) return DFTA_OK;                       // empty batch
    DFTA_REQUIRE(ctx, V && l && E && nV > 0, "null input");
    DFTA_REQUIRE(ctx, kind == DFTA_SWEEP_COUNT || kind == DFTA_SWEEP_ZERO, "kind");
    DFTA_REQUIRE(ctx, kind != DFTA_SWEEP_COUNT || (nodesLimit && count_out), "COUNT needs nodesLimit and count_out");
    DFTA_REQUIRE(ctx, kind != DFTA_SWEEP_ZERO || u0_out, "ZERO needs u0_out");
    if (kind == DFTA_SWEEP_COUNT)
        for (int t = 0; t < ntrials; ++t) DFTA_REQUIRE(ctx, nodesLimit[t] >= 0 && nodesLimit[t] < (1 << 30), "nodesLimit out of range");
    if (ntrials == 0) return DFTA_OK;
    const int N = g->N;
    Grouping G;
    if (make_grouping(ntrials, vidx, l, nV, G) != DFTA_OK) { snprintf(ctx->err, sizeof(ctx->err), "invalid vidx/l"); return DFTA_ERR_INVALID; }

    std::vector<double> sE(ntrials), sUs(ntrials), sUs1(ntrials);
    std::vector<int> sLim(ntrials, 0), sStart(ntrials);
    for (int s = 0; s < ntrials; ++s) {
        const int t = G.order[s];
        sE[s] = E[t];
        if (nodesLimit) sLim[s] = nodesLimit[t];
        if (boundary == DFTA_BOUNDARY_HOST) {
            if (g->uniform) host_boundary_uniform(g, E[t], static_cast<unsigned>(l[t]), false, &sStart[s], &sUs[s], &sUs1[s], nullptr);
            else host_boundary(g, E[t], &sStart[s], &sUs[s], &sUs1[s]);
        }
    }
    DevBuf<double> dV, dE, dUs, dUs1, dU0;
    DevBuf<int> dLim, dStart, dCount, dTrip, dSlotV, dSlotL, dBs, dBf, dBc;
    DevBuf<double2> dTab;
    hipStream_t st = ctx->stream;
    DFTA_HIP(ctx, dV.alloc((size_t)nV * N));
    DFTA_HIP(ctx, hipMemcpyAsync(dV.p, V, (size_t)nV * N * sizeof(double), hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, upload(dE, sE, st));
    DFTA_HIP(ctx, upload(dLim, sLim, st));
    DFTA_HIP(ctx, upload(dSlotV, G.slot_v, st));
    DFTA_HIP(ctx, upload(dSlotL, G.slot_l, st));
    DFTA_HIP(ctx, upload(dBs, G.blk_slot, st));
    DFTA_HIP(ctx, upload(dBf, G.blk_first, st));
    DFTA_HIP(ctx, upload(dBc, G.blk_cnt, st));
    DFTA_HIP(ctx, dTab.alloc(G.slot_v.size() * (size_t)N));
    DFTA_HIP(ctx, dCount.alloc(ntrials));
    DFTA_HIP(ctx, dTrip.alloc(ntrials));
    DFTA_HIP(ctx, dU0.alloc(ntrials));
    if (boundary == DFTA_BOUNDARY_HOST) {
        DFTA_HIP(ctx, upload(dStart, sStart, st));
        DFTA_HIP(ctx, upload(dUs, sUs, st));
        DFTA_HIP(ctx, upload(dUs1, sUs1, st));
    } else {
        DFTA_HIP(ctx, dStart.alloc(ntrials));
        DFTA_HIP(ctx, dUs.alloc(ntrials));
        DFTA_HIP(ctx, dUs1.alloc(ntrials));
        int rc = dfta_launch_boundary(ctx, g, dE.p, ntrials, dStart.p, dUs.p, dUs1.p);
        if (rc) return rc;
    }
    DevBuf<double2> dBounds;
    DFTA_HIP(ctx, dBounds.alloc(G.slot_v.size() * (size_t)dfta_bounds_stride(g)));
    int rc = dfta_launch_build_tab(ctx, g, dTab.p, dV.p, dSlotV.p, dSlotL.p, (int)G.slot_v.size(), dBounds.p);
    if (rc) return rc;
    DFTA_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    rc = dfta_launch_sweep(ctx, g, kind, nullptr, (int)G.blk_slot.size(), dTab.p, dBs.p, dBf.p, dBc.p, dE.p, dLim.p, dStart.p,
                           dUs.p, dUs1.p, dCount.p, dU0.p, dTrip.p, nullptr, dBounds.p, nullptr, nullptr, dSlotL.p);
    if (rc) return rc;
    DFTA_HIP(ctx, hipEventRecord(ctx->ev[1], st));
    ctx->have_kernel_time = true;
    std::vector<int> hCount(ntrials), hTrip(ntrials), hStart(ntrials);
    std::vector<double> hU0(ntrials);
    DFTA_HIP(ctx, hipMemcpyAsync(hCount.data(), dCount.p, ntrials * sizeof(int), hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipMemcpyAsync(hTrip.data(), dTrip.p, ntrials * sizeof(int), hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipMemcpyAsync(hStart.data(), dStart.p, ntrials * sizeof(int), hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipMemcpyAsync(hU0.data(), dU0.p, ntrials * sizeof(double), hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    for (int s = 0; s < ntrials; ++s) {
        const int t = G.order[s];
        if (count_out && kind == DFTA_SWEEP_COUNT) count_out[t] = hCount[s];
        if (u0_out) u0_out[t] = hU0[s];
        if (start_out) start_out[t] = hStart[s];
        if (trip_out) trip_out[t] = hTrip[s];
    }
    return DFTA_OK;
}

extern "C" int dfta_numerov_sweeps_dev(dfta_ctx* ctx, const dfta_grid* g, int kind, int nV, const double* dV, int ngroups,
                                       const int* group_off, const int* group_vidx, const int* group_l, const double* dE,
                                       const int* dLimit, const int* dStart, const double* dUs, const double* dUs1,
                                       int* dCount, double* dU0, int* dStartOut, int* dTrip)
{
    if (!ctx || !g) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, dV && group_off && group_vidx && group_l && dE && ngroups > 0, "null input");
    const int N = g->N;
    const int ntrials = group_off[ngroups];
    std::vector<int> slot_v(group_vidx, group_vidx + ngroups), slot_l(group_l, group_l + ngroups), bs, bf, bc;
    for (int k = 0; k < ngroups; ++k) {
        DFTA_REQUIRE(ctx, slot_v[k] >= 0 && slot_v[k] < nV && slot_l[k] >= 0 && slot_l[k] <= 3, "group vidx/l");
        for (int s = group_off[k]; s < group_off[k + 1]; s += 64) {
            bs.push_back(k); bf.push_back(s); bc.push_back(std::min(64, group_off[k + 1] - s));
        }
    }
    hipStream_t st = ctx->stream;
    DevBuf<int> dSlotV, dSlotL, dBs, dBf, dBc, dSt;
    DevBuf<double> dA, dB;
    DevBuf<double2> dTab;
    DFTA_HIP(ctx, upload(dSlotV, slot_v, st));
    DFTA_HIP(ctx, upload(dSlotL, slot_l, st));
    DFTA_HIP(ctx, upload(dBs, bs, st));
    DFTA_HIP(ctx, upload(dBf, bf, st));
    DFTA_HIP(ctx, upload(dBc, bc, st));
    DFTA_HIP(ctx, dTab.alloc((size_t)ngroups * N));
    const int* pStart = dStart;
    const double *pUs = dUs, *pUs1 = dUs1;
    if (!dStart || !dUs || !dUs1) {
        DFTA_HIP(ctx, dSt.alloc(ntrials));
        DFTA_HIP(ctx, dA.alloc(ntrials));
        DFTA_HIP(ctx, dB.alloc(ntrials));
        int rc = dfta_launch_boundary(ctx, g, dE, ntrials, dSt.p, dA.p, dB.p);
        if (rc) return rc;
        pStart = dSt.p; pUs = dA.p; pUs1 = dB.p;
    }
    DevBuf<double2> dBounds;
    DFTA_HIP(ctx, dBounds.alloc((size_t)ngroups * dfta_bounds_stride(g)));
    int rc = dfta_launch_build_tab(ctx, g, dTab.p, dV, dSlotV.p, dSlotL.p, ngroups, dBounds.p);
    if (rc) return rc;
    rc = dfta_launch_sweep(ctx, g, kind, nullptr, (int)bs.size(), dTab.p, dBs.p, dBf.p, dBc.p, dE, dLimit, pStart, pUs, pUs1, dCount,
                           dU0, dTrip, nullptr, dBounds.p, nullptr, nullptr, dSlotL.p);
    if (rc) return rc;
    if (dStartOut) DFTA_HIP(ctx, hipMemcpyAsync(dStartOut, pStart, ntrials * sizeof(int), hipMemcpyDeviceToDevice, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));   // scratch buffers die with this scope
    return DFTA_OK;
}

extern "C" int dfta_numerov_match(dfta_ctx* ctx, const dfta_grid* g, int boundary, int nV, const double* V, int ntrials,
                                  const int* vidx, const int* l, const double* E, double* Psi_out, long* matchPoint_out)
{
    if (!ctx || !g) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, V && l && E && Psi_out && matchPoint_out && nV > 0 && ntrials >= 0, "null input");
    if (ntrials == 0) return DFTA_OK;
    const int N = g->N;
    Grouping G;
    if (make_grouping(ntrials, vidx, l, nV, G) != DFTA_OK) { snprintf(ctx->err, sizeof(ctx->err), "invalid vidx/l"); return DFTA_ERR_INVALID; }
    std::vector<double> sE(ntrials), sUs(ntrials), sUs1(ntrials), sUz(ntrials, 0.0);
    std::vector<int> sStart(ntrials), sL(ntrials);
    for (int s = 0; s < ntrials; ++s) {
        const int t = G.order[s];
        sE[s] = E[t];
        sL[s] = l[t];
        if (boundary == DFTA_BOUNDARY_HOST) {
            if (g->uniform) host_boundary_uniform(g, E[t], static_cast<unsigned>(l[t]), true, &sStart[s], &sUs[s], &sUs1[s], &sUz[s]);
            else host_boundary(g, E[t], &sStart[s], &sUs[s], &sUs1[s]);
        }
    }
    hipStream_t st = ctx->stream;
    DevBuf<double> dV, dE, dUs, dUs1, dPsi, dQ, dUz;
    DevBuf<int> dStart, dSlotV, dSlotL, dTs, dL, dMp;
    DevBuf<double2> dTab;
    DFTA_HIP(ctx, dV.alloc((size_t)nV * N));
    DFTA_HIP(ctx, hipMemcpyAsync(dV.p, V, (size_t)nV * N * sizeof(double), hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, upload(dE, sE, st));
    DFTA_HIP(ctx, upload(dL, sL, st));
    DFTA_HIP(ctx, upload(dSlotV, G.slot_v, st));
    DFTA_HIP(ctx, upload(dSlotL, G.slot_l, st));
    DFTA_HIP(ctx, upload(dTs, G.trial_slot, st));
    DFTA_HIP(ctx, dTab.alloc(G.slot_v.size() * (size_t)N));
    DFTA_HIP(ctx, dPsi.alloc((size_t)ntrials * N));
    DFTA_HIP(ctx, dQ.alloc((size_t)ntrials * N));
    DFTA_HIP(ctx, dMp.alloc(ntrials));
    if (boundary == DFTA_BOUNDARY_HOST) {
        DFTA_HIP(ctx, upload(dStart, sStart, st));
        DFTA_HIP(ctx, upload(dUs, sUs, st));
        DFTA_HIP(ctx, upload(dUs1, sUs1, st));
        DFTA_HIP(ctx, upload(dUz, sUz, st));
    } else {
        DFTA_HIP(ctx, dStart.alloc(ntrials));
        DFTA_HIP(ctx, dUs.alloc(ntrials));
        DFTA_HIP(ctx, dUs1.alloc(ntrials));
        DFTA_HIP(ctx, dUz.alloc(ntrials));
        int rc = dfta_launch_boundary(ctx, g, dE.p, ntrials, dStart.p, dUs.p, dUs1.p, 1, dL.p, dUz.p);
        if (rc) return rc;
    }
    DevBuf<double2> dBounds;
    DFTA_HIP(ctx, dBounds.alloc(G.slot_v.size() * (size_t)dfta_bounds_stride(g)));
    int rc = dfta_launch_build_tab(ctx, g, dTab.p, dV.p, dSlotV.p, dSlotL.p, (int)G.slot_v.size(), dBounds.p);
    if (rc) return rc;
    rc = dfta_launch_match(ctx, g, ntrials, dTab.p, dTs.p, dE.p, dStart.p, dUs.p, dUs1.p, dL.p, dPsi.p, dQ.p, dMp.p, g->uniform ? nullptr : dBounds.p,
                           dUz.p);
    if (rc) return rc;
    std::vector<double> hPsi((size_t)ntrials * N);
    std::vector<int> hMp(ntrials);
    DFTA_HIP(ctx, hipMemcpyAsync(hPsi.data(), dPsi.p, hPsi.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipMemcpyAsync(hMp.data(), dMp.p, ntrials * sizeof(int), hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    for (int s = 0; s < ntrials; ++s) {
        const int t = G.order[s];
        memcpy(Psi_out + (size_t)t * N, hPsi.data() + (size_t)s * N, sizeof(double) * N);
        matchPoint_out[t] = hMp[s];
    }
    return DFTA_OK;
}

// ---- a potential resident on the device (include/dftatom_hip.h: dfta_potential) -------------------------------------------------
// The reference's Numerov holds a REFERENCE to the caller's Potential and re-reads it on every call (Numerov.h:69,186); its L3 makes
// ~2100 calls on one Numerov object per SCF step.  dfta_numerov_sweeps re-uploads the 1 MB potential and rebuilds the slot table on each
// of them (0.8 ... 6 ms per call); here both are done once per potential, and a call costs its sweep.
struct dfta_potential {
    dfta_ctx* ctx = nullptr;
    const dfta_grid* g = nullptr;
    std::vector<double> h_V;
    double* dV = nullptr;
    double2* dTab = nullptr;        // 4 slots: l = 0..3
    double2* dBounds = nullptr;
    int* dSlots = nullptr;          // slot_v (4 zeros), slot_l (0..3)
    dfta_scan_tables scan;          // tolerance mode: built on first use
    bool scan_built = false;
    // per-call scratch, grown on demand: one staging block in, one out
    size_t cap = 0;
    char* dIn = nullptr;
    char* dOut = nullptr;
    std::vector<char> hIn, hOut;
    double *dPsi = nullptr, *dQ = nullptr;
    size_t psi_cap = 0;
};

namespace {
int potential_build(dfta_potential* p)
{
    dfta_ctx* ctx = p->ctx;
    const int N = p->g->N;
    DFTA_HIP(ctx, hipMemcpyAsync(p->dV, p->h_V.data(), sizeof(double) * N, hipMemcpyHostToDevice, ctx->stream));
    int rc = dfta_launch_build_tab(ctx, p->g, p->dTab, p->dV, p->dSlots, p->dSlots + 4, 4, p->dBounds);
    if (rc) return rc;
    if (p->scan_built) rc = dfta_launch_scan_build_tab(ctx, p->g, p->scan, p->dV, p->dSlots, p->dSlots + 4);
    return rc;
}
int potential_scratch(dfta_potential* p, int ntrials)
{
    dfta_ctx* ctx = p->ctx;
    const size_t need = (size_t)std::max(ntrials, 64) * 64;      // 64 bytes per trial either way
    if (need <= p->cap) return DFTA_OK;
    if (p->dIn) (void)hipFree(p->dIn);
    if (p->dOut) (void)hipFree(p->dOut);
    p->dIn = p->dOut = nullptr; p->cap = 0;
    DFTA_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&p->dIn), need));
    DFTA_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&p->dOut), need));
    p->hIn.resize(need); p->hOut.resize(need);
    p->cap = need;
    return DFTA_OK;
}
}  // namespace

extern "C" void dfta_potential_destroy(dfta_potential* p)
{
    if (!p) return;
    for (void* q : {(void*)p->dV, (void*)p->dTab, (void*)p->dBounds, (void*)p->dSlots, (void*)p->dIn, (void*)p->dOut, (void*)p->dPsi, (void*)p->dQ}) if (q) (void)hipFree(q);
    dfta_scan_tables_destroy(&p->scan);
    delete p;
}

extern "C" int dfta_potential_create(dfta_ctx* ctx, const dfta_grid* g, const double* V, dfta_potential** out)
{
    if (!ctx || !g || !V || !out) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    const int N = g->N;
    dfta_potential* p = new dfta_potential();
    p->ctx = ctx; p->g = g;
    p->h_V.assign(V, V + N);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p->dV), sizeof(double) * N);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->dTab), sizeof(double2) * 4 * (size_t)N);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->dBounds), sizeof(double2) * 4 * (size_t)dfta_bounds_stride(g));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p->dSlots), sizeof(int) * 8);
    const int slots[8] = {0, 0, 0, 0, 0, 1, 2, 3};
    if (e == hipSuccess) e = hipMemcpy(p->dSlots, slots, sizeof(slots), hipMemcpyHostToDevice);
    if (e != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "dfta_potential_create: %s", hipGetErrorString(e)); dfta_potential_destroy(p); return DFTA_ERR_HIP; }
    const int rc = potential_build(p);
    if (rc) { dfta_potential_destroy(p); return rc; }
    *out = p;
    return DFTA_OK;
}

extern "C" int dfta_potential_update(dfta_potential* p, const double* V)
{
    if (!p || !V) return DFTA_ERR_INVALID;
    DFTA_ENTER(p->ctx);
    if (memcmp(V, p->h_V.data(), sizeof(double) * p->g->N) == 0) return DFTA_OK;      // what the reference would re-read is what is resident
    DFTA_HIP(p->ctx, hipStreamSynchronize(p->ctx->stream));                            // h_V may still be the source of a copy
    p->h_V.assign(V, V + p->g->N);
    return potential_build(p);
}

extern "C" int dfta_potential_sweeps(dfta_potential* p, int kind, int sweep_mode, int ntrials, const int* l, const double* E, const int* nodesLimit,
                                     int* count_out, double* u0_out, int* start_out, int* trip_out)
{
    if (!p) return DFTA_ERR_INVALID;
    dfta_ctx* ctx = p->ctx;
    const dfta_grid* g = p->g;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, l && E && ntrials >= 0, "null input");
    DFTA_REQUIRE(ctx, kind == DFTA_SWEEP_COUNT || kind == DFTA_SWEEP_ZERO, "kind");
    DFTA_REQUIRE(ctx, kind != DFTA_SWEEP_COUNT || (nodesLimit && count_out), "COUNT needs nodesLimit and count_out");
    DFTA_REQUIRE(ctx, kind != DFTA_SWEEP_ZERO || u0_out, "ZERO needs u0_out");
    DFTA_REQUIRE(ctx, sweep_mode == DFTA_SWEEPS_EXACT || (sweep_mode == DFTA_SWEEPS_TOLERANCE && dfta_scan_supported(g)), "sweep mode / grid");
    if (ntrials == 0) return DFTA_OK;
    hipStream_t st = ctx->stream;
    int rc = potential_scratch(p, ntrials);
    if (rc) return rc;
    // staging block in: E, us, us1 (doubles), then limit, start, blk_slot, blk_first, blk_cnt / trial_slot (ints): one copy each way
    const size_t nt = ntrials;
    double* hE = reinterpret_cast<double*>(p->hIn.data());
    double *hUs = hE + nt, *hUs1 = hUs + nt;
    int* hLim = reinterpret_cast<int*>(hUs1 + nt);
    int *hStart = hLim + nt, *hBs = hStart + nt, *hBf = hBs + nt, *hBc = hBf + nt;
    double* dE = reinterpret_cast<double*>(p->dIn);
    double *dUs = dE + nt, *dUs1 = dUs + nt;
    int* dLim = reinterpret_cast<int*>(dUs1 + nt);
    int *dStart = dLim + nt, *dBs = dStart + nt, *dBf = dBs + nt, *dBc = dBf + nt;
    double* dU0 = reinterpret_cast<double*>(p->dOut);
    int* dCount = reinterpret_cast<int*>(dU0 + nt);
    int *dTrip = dCount + nt, *dStartOut = dTrip + nt, *dBad = dStartOut + nt;
    if (sweep_mode == DFTA_SWEEPS_TOLERANCE) {
        if (!p->scan_built) {
            rc = dfta_scan_tables_create(ctx, g, 4, &p->scan);
            if (rc) return rc;
            p->scan_built = true;
            rc = dfta_launch_scan_build_tab(ctx, g, p->scan, p->dV, p->dSlots, p->dSlots + 4);
            if (rc) return rc;
        }
        for (size_t t = 0; t < nt; ++t) {
            DFTA_REQUIRE(ctx, l[t] >= 0 && l[t] <= 3, "l");
            hE[t] = E[t]; hLim[t] = nodesLimit ? nodesLimit[t] : 0; hBs[t] = l[t];
        }
        DFTA_HIP(ctx, hipMemcpyAsync(p->dIn, p->hIn.data(), nt * 64, hipMemcpyHostToDevice, st));
        DFTA_HIP(ctx, hipEventRecord(ctx->ev[0], st));
        rc = dfta_launch_scan_sweeps(ctx, g, kind, ntrials, p->scan, dBs, dE, dLim, dCount, dU0, dStartOut, dTrip, dBad);
        if (rc) return rc;
        DFTA_HIP(ctx, hipEventRecord(ctx->ev[1], st));
        ctx->have_kernel_time = true;
        DFTA_HIP(ctx, hipMemcpyAsync(p->hOut.data(), p->dOut, nt * 64, hipMemcpyDeviceToHost, st));
        DFTA_HIP(ctx, hipStreamSynchronize(st));
        const double* oU0 = reinterpret_cast<const double*>(p->hOut.data());
        const int* oCount = reinterpret_cast<const int*>(oU0 + nt);
        const int *oTrip = oCount + nt, *oStart = oTrip + nt, *oBad = oStart + nt;
        bool anybad = false;
        for (size_t t = 0; t < nt; ++t) {
            anybad = anybad || oBad[t];
            if (count_out && kind == DFTA_SWEEP_COUNT) count_out[t] = oCount[t];
            if (u0_out) u0_out[t] = oU0[t];
            if (start_out) start_out[t] = oStart[t];
            if (trip_out) trip_out[t] = oTrip[t];
        }
        if (!anybad) return DFTA_OK;
        // a trial the scan could not decide: the whole call again on the exact kernels
    }
    Grouping G;
    if (make_grouping(ntrials, nullptr, l, 1, G) != DFTA_OK) { snprintf(ctx->err, sizeof(ctx->err), "invalid l"); return DFTA_ERR_INVALID; }
    for (size_t s = 0; s < nt; ++s) {
        const int t = G.order[s];
        hE[s] = E[t];
        hLim[s] = nodesLimit ? nodesLimit[t] : 0;
        if (g->uniform) host_boundary_uniform(g, E[t], static_cast<unsigned>(l[t]), false, &hStart[s], &hUs[s], &hUs1[s], nullptr);
        else host_boundary(g, E[t], &hStart[s], &hUs[s], &hUs1[s]);
    }
    const size_t nb = G.blk_slot.size();
    for (size_t b = 0; b < nb; ++b) { hBs[b] = G.slot_l[G.blk_slot[b]]; hBf[b] = G.blk_first[b]; hBc[b] = G.blk_cnt[b]; }    // table slot = l
    DFTA_HIP(ctx, hipMemcpyAsync(p->dIn, p->hIn.data(), nt * 64, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    rc = dfta_launch_sweep(ctx, g, kind, nullptr, (int)nb, p->dTab, dBs, dBf, dBc, dE, dLim, dStart, dUs, dUs1, dCount, dU0, dTrip, nullptr, p->dBounds, nullptr,
                           nullptr, p->dSlots + 4);
    if (rc) return rc;
    DFTA_HIP(ctx, hipEventRecord(ctx->ev[1], st));
    ctx->have_kernel_time = true;
    DFTA_HIP(ctx, hipMemcpyAsync(p->hOut.data(), p->dOut, nt * 64, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    const double* oU0 = reinterpret_cast<const double*>(p->hOut.data());
    const int* oCount = reinterpret_cast<const int*>(oU0 + nt);
    const int* oTrip = oCount + nt;
    for (size_t s = 0; s < nt; ++s) {
        const int t = G.order[s];
        if (count_out && kind == DFTA_SWEEP_COUNT) count_out[t] = oCount[s];
        if (u0_out) u0_out[t] = oU0[s];
        if (start_out) start_out[t] = hStart[s];
        if (trip_out) trip_out[t] = oTrip[s];
    }
    return DFTA_OK;
}

extern "C" int dfta_potential_match(dfta_potential* p, int ntrials, const int* l, const double* E, double* Psi_out, long* matchPoint_out)
{
    if (!p) return DFTA_ERR_INVALID;
    dfta_ctx* ctx = p->ctx;
    const dfta_grid* g = p->g;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, l && E && Psi_out && matchPoint_out && ntrials >= 0, "null input");
    if (ntrials == 0) return DFTA_OK;
    const int N = g->N;
    hipStream_t st = ctx->stream;
    int rc = potential_scratch(p, ntrials);
    if (rc) return rc;
    if ((size_t)ntrials > p->psi_cap) {
        if (p->dPsi) (void)hipFree(p->dPsi);
        if (p->dQ) (void)hipFree(p->dQ);
        p->dPsi = p->dQ = nullptr; p->psi_cap = 0;
        DFTA_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&p->dPsi), sizeof(double) * (size_t)ntrials * N));
        DFTA_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&p->dQ), sizeof(double) * (size_t)ntrials * N));
        p->psi_cap = ntrials;
    }
    const size_t nt = ntrials;
    double* hE = reinterpret_cast<double*>(p->hIn.data());
    double *hUs = hE + nt, *hUs1 = hUs + nt, *hUz = hUs1 + nt;
    int* hStart = reinterpret_cast<int*>(hUz + nt);
    int *hL = hStart + nt, *hTs = hL + nt;
    double* dE = reinterpret_cast<double*>(p->dIn);
    double *dUs = dE + nt, *dUs1 = dUs + nt, *dUz = dUs1 + nt;
    int* dStart = reinterpret_cast<int*>(dUz + nt);
    int *dL = dStart + nt, *dTs = dL + nt;
    int* dMp = reinterpret_cast<int*>(p->dOut);
    for (size_t t = 0; t < nt; ++t) {
        DFTA_REQUIRE(ctx, l[t] >= 0 && l[t] <= 3, "l");
        hE[t] = E[t]; hL[t] = l[t]; hTs[t] = l[t]; hUz[t] = 0;
        if (g->uniform) host_boundary_uniform(g, E[t], static_cast<unsigned>(l[t]), true, &hStart[t], &hUs[t], &hUs1[t], &hUz[t]);
        else host_boundary(g, E[t], &hStart[t], &hUs[t], &hUs1[t]);
    }
    DFTA_HIP(ctx, hipMemcpyAsync(p->dIn, p->hIn.data(), nt * 64, hipMemcpyHostToDevice, st));
    rc = dfta_launch_match(ctx, g, ntrials, p->dTab, dTs, dE, dStart, dUs, dUs1, dL, p->dPsi, p->dQ, dMp, g->uniform ? nullptr : p->dBounds, dUz);
    if (rc) return rc;
    DFTA_HIP(ctx, hipMemcpyAsync(Psi_out, p->dPsi, sizeof(double) * nt * N, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipMemcpyAsync(p->hOut.data(), dMp, sizeof(int) * nt, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    for (size_t t = 0; t < nt; ++t) matchPoint_out[t] = reinterpret_cast<const int*>(p->hOut.data())[t];
    return DFTA_OK;
}

#ifdef DFTA_PIPE_PROF
extern "C" int dfta_debug_pipe_prof(unsigned long long* out)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pipe_prof), sizeof(g_pipe_prof)) != hipSuccess) return 1;
    unsigned long long z[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pipe_prof), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif
