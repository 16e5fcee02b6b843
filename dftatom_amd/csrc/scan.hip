// scan.hip -- TOLERANCE MODE of the Numerov sweeps for gfx950: the three-term recurrence as a transfer-matrix scan.
//
// Replaces (opt-in: DFTA_SWEEP_MODE_TOLERANCE) the same reference functions as numerov.hip / levels.hip --
// Numerov<NonUniform>::SolveSchrodingerCountNodes / SolutionInZero / MatchSolutionCompletely (Numerov.h:272-504) and
// DFTAtom::LocateInterval / LoopOverLevels (DFTAtom.cpp:493-604) -- with a different ORDER OF ROUNDINGS (hence "tolerance": node
// counts and decisions agree with the exact kernels except inside the round-off band of a transition, eigenvalues to ~1e-11 |E|).
//
// Idea.  With u_i = w_i / (1 - f_i/12) the reference's recurrence (Numerov.h:309-321,510-513)
//     w_{i-1} = 2 w_i - w_{i+1} + u_i f_i
// is LINEAR in w:  w_{i-1} = (2 + g_i) w_i - w_{i+1},  g_i = f_i / (1 - f_i/12).  In the summed form
//     D_{i-1} = D_i + g_i w_i,   w_{i-1} = w_i + D_{i-1}          (D_i = w_i - w_{i+1})
// one step is the 2x2 matrix [[1+g, 1], [g, 1]] acting on (w, D); products of such matrices are associative, so the sweep of ONE
// trial is spread over the 1024 lanes of a workgroup: lane t multiplies the matrices of its own C = (N-1)/1024 grid points (two
// independent columns: 2 fma + 2 add per point, g from the table row in 8 instructions), a log-depth scan over the lanes combines
// the 1024 segment matrices (wave shuffles + one hand-over through LDS), and -- CountNodes only -- a second pass over the segment
// with the now known incoming (w, D) counts the sign changes.  A 131 073-point sweep takes ~25 us on one compute unit instead of
// 4 ms as a dependent chain, so the bisections of a level need no speculation: ONE workgroup runs LocateInterval and the
// u(0) bisection of its level from start to end on the device (k_scan_levels), ~150 sweeps back to back, no host round trips.
// The summed form carries the slope D separately (no 2w - w' cancellation), which makes it slightly MORE accurate than the
// reference's own rounding sequence; what differs from the exact kernels is only which way round-off falls.
//
// Table: per slot (potential, l) rows { veff_i, A_i = 2 Rp^2 delta^2 exp(2 i delta) }, f_i = A_i (veff_i - E) + delta^2/4
// (Numerov.h:96-101), LANE-INTERLEAVED: row i = t C + k is stored at k 1024 + t, so that the 1024 lanes read consecutive
// addresses at every step (the layout idea of the multigrid levels).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "internal.h"
#include "levels.h"

namespace {

constexpr int kT = 1024;          // lanes per trial = threads per workgroup
constexpr int kW = kT / 64;       // waves per workgroup
constexpr double kInv12 = 1. / 12.;

struct ScanGrid {
    int N, logC;                  // N - 1 = 1024 << logC
    double delta, far_thr, c4;    // delta, exp(a) < 1e-200 <=> a < far_thr, delta^2/4
    const double* r;              // r_i
};

struct Mat { double a, b, c, d; };   // [[a, b], [c, d]] acting on (w, D)

__device__ __forceinline__ Mat mat_mul(const Mat& L, const Mat& E)    // L after E
{
    Mat r;
    r.a = fma(L.a, E.a, L.b * E.c);
    r.b = fma(L.a, E.b, L.b * E.d);
    r.c = fma(L.c, E.a, L.d * E.c);
    r.d = fma(L.c, E.b, L.d * E.d);
    return r;
}

__device__ __forceinline__ double shfl_up_d(double v, int off) { return __shfl_up(v, off, 64); }

// g = f / (1 - f/12) = f (1 + x)(1 + x^2)(1 + x^4) + O(x^8), x = f/12.  |x| <= 2^-7 on every grid of BASELINE (f <= (460 delta)^2 at
// the cut-off) makes the remainder < 2^-56; rows beyond that (large delta; the innermost rows of l = 3, where f -> l(l+1)) are
// detected per sweep (xmax) and the sweep is repeated with an IEEE division per row
template <bool DIV>
__device__ __forceinline__ double g_of(double f, double& xmax)
{
    const double x = f * kInv12;
    xmax = fmax(xmax, fabs(x));
    if (DIV) return f / (1. - x);
    const double x2 = x * x;
    const double t1 = fma(f, x, f);
    const double t2 = fma(t1, x2, t1);
    return fma(t2, x2 * x2, t2);
}
constexpr double kSeriesMax = 0.0078125;

// turning-point summary of a stretch of rows in sweep order (descending index); combine(X earlier/higher, Y later/lower)
struct Turn { int amax, fmax, fbelow; };   // largest allowed index (veff <= E), largest forbidden index, largest forbidden index below amax
__device__ __forceinline__ Turn turn_combine(const Turn& X, const Turn& Y)
{
    Turn r;
    r.amax = X.amax >= 0 ? X.amax : Y.amax;
    r.fmax = X.fmax >= 0 ? X.fmax : Y.fmax;
    r.fbelow = X.fbelow >= 0 ? X.fbelow : (X.amax >= 0 ? Y.fmax : Y.fbelow);
    return r;
}

struct ScanShared {
    Mat wave_tot[kW];
    Turn wave_turn[kW];
    int red[kW];
};

struct SweepOut {
    int count;          // COUNT: min(sign changes, limit + 1) (+ the final extrapolated test); the decision value of CountNodes
    int start, iexit;   // cut-off index, index at which CountNodes returned (0: ran to the end)
    double u0;          // ZERO: u_1 (2 + f_1) - u_2 (Numerov.h:398)
    double w1, w2;      // the last two w of the sweep (indices 1, 2) when it ran to the end
    int bad;            // a non-finite value or f >= 12 above the innermost row: the exact kernels must decide this trial
};

__device__ __forceinline__ int block_min_int(int v, ScanShared& sh, int tid)
{
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) sh.red[tid >> 6] = v;
    __syncthreads();
    int m = sh.red[0];
#pragma unroll
    for (int w = 1; w < kW; ++w) m = min(m, sh.red[w]);
    return m;
}
__device__ __forceinline__ int block_sum_int(int v, ScanShared& sh, int tid)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) sh.red[tid >> 6] = v;
    __syncthreads();
    int m = 0;
#pragma unroll
    for (int w = 0; w < kW; ++w) m += sh.red[w];
    return m;
}

__device__ __forceinline__ double far_arg_s(const double* __restrict__ r, int i, double s, double delta)
{
    return -r[i] * s - static_cast<double>(i) * delta * 0.5;      // Numerov.h:107
}

// GetMaxRadiusIndex (Numerov.h:119-136) for a monotone start value: the smallest index >= 2 whose start value is below 1e-200, N-1 if none
__device__ int scan_cutoff(const ScanGrid& G, double s, ScanShared& sh, int tid)
{
    const int C = 1 << G.logC;
    // first the segment: lane tid looks at the LAST index of segment tid
    const int iend = min(tid * C + C, G.N - 1);
    int cand = (iend >= 2 && far_arg_s(G.r, iend, s, G.delta) < G.far_thr) ? tid : kT;
    const int seg = block_min_int(cand, sh, tid);
    if (seg >= kT) return G.N - 1;
    // then the index inside (seg C, seg C + C]
    int best = G.N;
    for (int k = tid; k < C; k += kT) {
        const int i = seg * C + 1 + k;
        if (i >= 2 && far_arg_s(G.r, i, s, G.delta) < G.far_thr) { best = i; break; }     // ascending k: the first hit of this lane
    }
    return block_min_int(best, sh, tid);
}

constexpr int kBatch = 8;
// rows [klo, khi] of a lane's segment (descending), kBatch table rows in flight
template <typename F>
__device__ __forceinline__ void for_rows(const double2* __restrict__ p, int khi, int klo, F&& step)
{
    int k = khi;
    for (; k - (kBatch - 1) >= klo; k -= kBatch) {
        double2 r[kBatch];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) r[j] = p[(size_t)(k - j) << 10];
#pragma unroll
        for (int j = 0; j < kBatch; ++j) step(r[j]);
    }
    for (; k >= klo; --k) step(p[(size_t)k << 10]);
}

// the innermost rows (index < 16) are always divided: there f -> l(l+1)/i^2 is far outside the series' range
constexpr int kInnerRows = 16;

// pass 1: the transfer matrix of rows [klo, khi] (two columns), min of f (all rows forbidden <=> fmin >= 0), max |f/12|
template <bool DIV>
__device__ __forceinline__ void pass1_rows(const double2* __restrict__ p, int khi, int klo, bool inner, double E, double c4, Mat& M, double& fmin, double& xmax)
{
    auto step_s = [&](const double2 row) {
        const double f = fma(row.y, row.x - E, c4);
        const double g = g_of<DIV>(f, xmax);
        fmin = fmin < f ? fmin : f;
        M.c = fma(g, M.a, M.c); M.a += M.c;
        M.d = fma(g, M.b, M.d); M.b += M.d;
    };
    if (!inner || DIV) { for_rows(p, khi, klo, step_s); return; }
    double xin = 0;
    auto step_d = [&](const double2 row) {
        const double f = fma(row.y, row.x - E, c4);
        const double g = g_of<true>(f, xin);
        fmin = fmin < f ? fmin : f;
        M.c = fma(g, M.a, M.c); M.a += M.c;
        M.d = fma(g, M.b, M.d); M.b += M.d;
    };
    if (khi >= kInnerRows) for_rows(p, khi, max(klo, kInnerRows), step_s);
    if (klo < kInnerRows) for_rows(p, min(khi, kInnerRows - 1), klo, step_d);
}

// pass 2: sign changes between consecutive w over rows [klo, khi], starting from (w, D)
template <bool DIV>
__device__ __forceinline__ int pass2_rows(const double2* __restrict__ p, int khi, int klo, bool inner, double E, double c4, double w, double D)
{
    int cnt = 0;
    double xm = 0;
    auto step_s = [&](const double2 row) {
        const double f = fma(row.y, row.x - E, c4);
        const double g = g_of<DIV>(f, xm);
        D = fma(g, w, D);
        const double wn = w + D;
        cnt += ((wn > 0.) != (w > 0.));
        w = wn;
    };
    if (!inner || DIV) { for_rows(p, khi, klo, step_s); return cnt; }
    auto step_d = [&](const double2 row) {
        const double f = fma(row.y, row.x - E, c4);
        const double g = g_of<true>(f, xm);
        D = fma(g, w, D);
        const double wn = w + D;
        cnt += ((wn > 0.) != (w > 0.));
        w = wn;
    };
    if (khi >= kInnerRows) for_rows(p, khi, max(klo, kInnerRows), step_s);
    if (klo < kInnerRows) for_rows(p, min(khi, kInnerRows - 1), klo, step_d);
    return cnt;
}

// One sweep of one trial by the whole workgroup.  KIND: DFTA_SWEEP_COUNT / DFTA_SWEEP_ZERO.  tab: the slot's interleaved table; mm: per
// lane {min, max} of veff over the lane's rows ({-inf, +inf} when a row is NaN).
template <int KIND>
__device__ SweepOut scan_sweep(const ScanGrid& G, const double2* __restrict__ tab, const double2* __restrict__ mm, double E, int limit, ScanShared& sh)
{
    const int tid = threadIdx.x;
    const int t = kT - 1 - tid;                 // segment of this lane: thread order = sweep order (descending index)
    const int C = 1 << G.logC;
    const int lane = tid & 63, wv = tid >> 6;
    SweepOut out;
    out.count = 0; out.u0 = 0; out.bad = 0; out.iexit = 0; out.w1 = out.w2 = 0;
    const double sq = sqrt(2. * fabs(E));
    const int s = scan_cutoff(G, sq, sh, tid);
    out.start = s;
    const double us = exp(far_arg_s(G.r, s, sq, G.delta));          // GetBoundaryValueFar at the cut-off and one point inside
    const double us1 = exp(far_arg_s(G.r, s - 1, sq, G.delta));
    auto row_of = [&](int i) -> double2 { return (i == (kT << G.logC)) ? tab[(size_t)kT << G.logC] : tab[((size_t)(i & (C - 1)) << 10) + (i >> G.logC)]; };
    const double2 rs = row_of(s), rs1 = row_of(s - 1);
    const double fs = fma(rs.y, rs.x - E, G.c4), fs1 = fma(rs1.y, rs1.x - E, G.c4);
    const double ws = (1. - kInv12 * fs) * us, ws1 = (1. - kInv12 * fs1) * us1;        // Numerov.h:297,302
    const int ibase = t << G.logC;
    const double2* p = tab + t;
    // ---- CountNodes: where does the sweep leave through the classical turning point (Numerov.h:337-341)?  The loop index runs over
    // [1, s-2]; lanes whose rows lie on one side of E answer from {min, max}, the others look at their rows
    int iexit = 0;
    if (KIND == DFTA_SWEEP_COUNT) {
        Turn tu = {-1, -1, -1};
        const int top = min(s - 2, ibase + C - 1), bot = max(1, ibase);
        if (top >= bot) {
            const double2 m = mm[t];
            if (m.x > E) tu.fmax = top;                 // every row forbidden
            else if (m.y <= E) tu.amax = top;           // every row allowed
            else {
                int i = top;
                for_rows(p, top - ibase, bot - ibase, [&](const double2 row) {
                    const double tt = row.x - E;
                    const bool al = tt <= 0., fo = tt > 0.;
                    tu.amax = (al && tu.amax < 0) ? i : tu.amax;
                    tu.fbelow = (fo && tu.amax >= 0 && tu.fbelow < 0) ? i : tu.fbelow;
                    tu.fmax = (fo && tu.fmax < 0) ? i : tu.fmax;
                    --i;
                });
            }
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {     // inclusive scan of the summaries: lane 63 ends up with the wave's
            Turn X;
            X.amax = __shfl_up(tu.amax, off, 64); X.fmax = __shfl_up(tu.fmax, off, 64); X.fbelow = __shfl_up(tu.fbelow, off, 64);
            if (lane >= off) tu = turn_combine(X, tu);
        }
        __syncthreads();
        if (lane == 63) sh.wave_turn[wv] = tu;
        __syncthreads();
        Turn all = sh.wave_turn[0];
#pragma unroll
        for (int q = 1; q < kW; ++q) all = turn_combine(all, sh.wave_turn[q]);
        iexit = all.fbelow >= 0 ? all.fbelow : 0;       // the first forbidden index below the first allowed one; 0: none
    }
    out.iexit = iexit;
    // ---- pass 1: the segment's transfer matrix (two columns) over its step rows in [lo, s-1]; step i maps (w_i, D_i) to (w_{i-1}, D_{i-1})
    const int lo = max(iexit + 1, 2), hi = s - 1;
    const int khi = min(hi - ibase, C - 1), klo = max(lo - ibase, 0);
    Mat M = {1., 0., 0., 1.};
    double fmin = 1., xmax = 0.;
    const bool inner = (t == 0);                     // the lane that owns the innermost rows
    if (khi >= klo) pass1_rows<false>(p, khi, klo, inner, E, G.c4, M, fmin, xmax);
    const bool use_div = __syncthreads_or(xmax > kSeriesMax) != 0;
    double xall = xmax;
    if (use_div) {                                   // a row outside the series' range somewhere: all lanes again, with the division
        M.a = 1.; M.b = 0.; M.c = 0.; M.d = 1.;
        fmin = 1.; xall = 0.;
        if (khi >= klo) pass1_rows<true>(p, khi, klo, inner, E, G.c4, M, fmin, xall);
    }
    // f >= 12 in a step row (d <= 0: u and w differ in sign) is left to the exact kernels.  (The series rows have |x| <= 2^-7, the
    // inner rows of pass1_rows<false> keep their own maximum: row 1 of l = 3 has f > 12 by construction and is no step row.)
    int bad = !(xall < 1.);
    // ---- combine: inclusive scan of the matrices in thread order (wave shuffles, then the 16 wave totals through LDS)
    Mat P = M;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Mat Ee;
        Ee.a = shfl_up_d(P.a, off); Ee.b = shfl_up_d(P.b, off); Ee.c = shfl_up_d(P.c, off); Ee.d = shfl_up_d(P.d, off);
        if (lane >= off) P = mat_mul(P, Ee);
    }
    if (lane == 63) sh.wave_tot[wv] = P;
    __syncthreads();
    // incoming state of this wave, and the final state of the sweep
    double w_in = ws1, D_in = ws1 - ws;
    double w_fin, D_fin;
    {
        double w = ws1, D = ws1 - ws;
#pragma unroll
        for (int q = 0; q < kW; ++q) {
            if (q == wv) { w_in = w; D_in = D; }
            const Mat Wq = sh.wave_tot[q];
            const double w2 = fma(Wq.a, w, Wq.b * D), D2 = fma(Wq.c, w, Wq.d * D);
            w = w2; D = D2;
        }
        w_fin = w; D_fin = D;
    }
    {   // exclusive prefix inside the wave
        Mat X;
        X.a = shfl_up_d(P.a, 1); X.b = shfl_up_d(P.b, 1); X.c = shfl_up_d(P.c, 1); X.d = shfl_up_d(P.d, 1);
        if (lane == 0) { X.a = 1; X.b = 0; X.c = 0; X.d = 1; }
        const double w2 = fma(X.a, w_in, X.b * D_in), D2 = fma(X.c, w_in, X.d * D_in);
        w_in = w2; D_in = D2;
    }
    bad |= !(fabs(w_fin) < INFINITY) || !(fabs(D_fin) < INFINITY);
    // u_1, u_2 and the extrapolation to the origin (Numerov.h:343-346,398) when the sweep ran to the end: the final state is
    // (w_1, w_1 - w_2).  Row 1 of l = 3 has f > 12 (d < 0): these last values are divided exactly
    double u1 = 0, u2 = 0;
    if (iexit <= 1) {
        const double2 r1 = row_of(1), r2 = row_of(2);
        const double f1 = fma(r1.y, r1.x - E, G.c4), f2 = fma(r2.y, r2.x - E, G.c4);
        out.w1 = w_fin; out.w2 = w_fin - D_fin;
        u1 = out.w1 / (1. - kInv12 * f1); u2 = out.w2 / (1. - kInv12 * f2);
        out.u0 = u1 * (2. + f1) - u2;
    }
    if (KIND == DFTA_SWEEP_COUNT) {
        // ---- pass 2: sign changes between w_i and w_{i-1} for the step rows i in [lo2, s-1] (u has the sign of w there: d > 0; the
        // comparison of step 2, u_2 against u_1, is taken from the divided values below).  A lane whose rows are all forbidden (g >= 0:
        // w'' = g w keeps |w| convex) crosses zero at most once: its count is the sign change between its two ends, known from M.
        const int lo2 = max(lo, 3);
        const int klo2 = max(lo2 - ibase, 0);
        int cnt = 0;
        if (khi >= klo2) {
            if (fmin >= 0. && klo2 == klo) {
                const double w_out = fma(M.a, w_in, M.b * D_in);
                cnt = (w_out > 0.) != (w_in > 0.);
            } else {
                cnt = use_div ? pass2_rows<true>(p, khi, klo2, inner, E, G.c4, w_in, D_in) : pass2_rows<false>(p, khi, klo2, inner, E, G.c4, w_in, D_in);
            }
        }
        int total = block_sum_int(cnt + (bad << 24), sh, tid);
        bad = (total >> 24) != 0;
        total &= (1 << 24) - 1;
        if (iexit <= 1 && s - 1 >= 2) total += ((u1 > 0.) != (u2 > 0.));       // step 2
        // the first comparison of the loop is against the sign of the start value u_{s-1} = us1 > 0 -- that of w_{s-1}, the incoming w
        // of the first lane.  Early return at count > limit: the value is limit + 1.
        if (total > limit) total = limit + 1;
        else if (iexit == 0) total += ((out.u0 > 0.) != (u1 > 0.));      // ran to the end: the extrapolated point (Numerov.h:343-347)
        out.count = total;
    } else {
        bad = __syncthreads_or(bad);
    }
    out.bad = bad;
    return out;
}

// per slot and lane: {min, max} of veff over the lane's rows
__global__ void __launch_bounds__(kT) k_scan_lane_minmax(const double2* __restrict__ tabs, int N, int logC, double2* __restrict__ mm)
{
    const size_t rows = (size_t)N;
    const double2* p = tabs + blockIdx.x * rows + threadIdx.x;
    const int C = 1 << logC;
    double lo = INFINITY, hi = -INFINITY;
    bool nan = false;
    for (int k = 0; k < C; ++k) {
        const double v = p[(size_t)k << 10].x;
        nan = nan || (v != v);
        lo = fmin(lo, v); hi = fmax(hi, v);
    }
    double2 o;
    o.x = nan ? -INFINITY : lo; o.y = nan ? INFINITY : hi;
    mm[(size_t)blockIdx.x * kT + threadIdx.x] = o;
}

__global__ void __launch_bounds__(kT) k_scan_sweeps(ScanGrid G, int kind, const double2* __restrict__ tabs, const double2* __restrict__ mms, const int* __restrict__ trial_slot,
                                                     const double* __restrict__ E, const int* __restrict__ limit, int* __restrict__ count,
                                                     double* __restrict__ u0, int* __restrict__ start, int* __restrict__ trip, int* __restrict__ bad)
{
    __shared__ ScanShared sh;
    const int q = blockIdx.x;
    const size_t rows = ((size_t)kT << G.logC) + 1;
    const double2* tab = tabs + (size_t)trial_slot[q] * rows;
    const double2* mm = mms + (size_t)trial_slot[q] * kT;
    SweepOut o;
    if (kind == DFTA_SWEEP_COUNT) o = scan_sweep<DFTA_SWEEP_COUNT>(G, tab, mm, E[q], limit[q], sh);
    else o = scan_sweep<DFTA_SWEEP_ZERO>(G, tab, mm, E[q], 0, sh);
    if (threadIdx.x == 0) {
        if (count) count[q] = o.count;
        if (u0) u0[q] = o.u0;
        if (start) start[q] = o.start;
        if (trip) trip[q] = o.start - 2 - (kind == DFTA_SWEEP_COUNT && o.iexit > 0 ? o.iexit - 1 : 0);
        if (bad) bad[q] = o.bad;
    }
}

// ---- LocateInterval + the u(0) bisection of one level (DFTAtom.cpp:493-604) by ONE workgroup, start to end on the device -----------
// Same three bisections, same midpoints (toe + boe) / 2, same predicates and stop rules as the reference; every trial is a scan sweep.
// chained != 0: the jobs of chain c run one after the other and hand E - 3 on as the next BottomEnergy (DFTAtom.cpp:541); otherwise a
// chain is one job that starts from its own (clamped) bottom.  Results go to the Job records exactly as k_walk leaves them.
constexpr double kErr = 1e-12;        // energyErr, DFTAtom.cpp:349
constexpr int kIter3 = 500;           // DFTAtom.cpp:517
constexpr int kPhDone = 4;            // PH_DONE of levels.hip

__global__ void __launch_bounds__(kT) k_scan_levels(ScanGrid G, dfta::Job* __restrict__ jobs, const int* __restrict__ chain_off, int chained,
                                                     const double2* __restrict__ tabs, const double2* __restrict__ mms, int fixed_point,
                                                     unsigned long long* __restrict__ counters)
{
    __shared__ ScanShared sh;
    const int c = blockIdx.x;
    const size_t rows = (size_t)G.N;
    double handed = 0;
    bool have_handed = false;
    for (int k = chain_off[c]; k < chain_off[c + 1]; ++k) {
        dfta::Job* J = jobs + k;
        if (J->frozen) continue;
        const int nodes = J->nodes, slot = J->slot;
        const double2* tab = tabs + (size_t)slot * rows;
        const double2* mm = mms + (size_t)slot * kT;
        const double bottom0 = (chained && have_handed) ? handed : J->bottom0;
        int n_count = 0, n_zero = 0, bad = 0, len2 = 0, n_fixed = 0;
        long long pts = 0;
        // first bisection: the lowest energy with more than `nodes` nodes (DFTAtom.cpp:568-585)
        double toe = 50., boe = bottom0;
        while (toe - boe > kErr) {
            const double E = (toe + boe) / 2;
            const SweepOut o = scan_sweep<DFTA_SWEEP_COUNT>(G, tab, mm, E, nodes, sh);
            ++n_count; bad |= o.bad;
            pts += o.start - 1 - (o.iexit > 0 ? o.iexit : 1);
            if (o.count > nodes) toe = E; else boe = E;
        }
        const double top = toe;
        // second bisection: the lowest energy with at least `nodes` nodes (DFTAtom.cpp:587-603).  "count < 0" never holds: the
        // path of a node-less level is arithmetic (counted, not integrated -- as in levels.hip)
        boe = bottom0;
        while (toe - boe > kErr) {
            const double E = (toe + boe) / 2;
            ++n_count; ++len2;
            if (nodes == 0) { toe = E; continue; }
            const SweepOut o = scan_sweep<DFTA_SWEEP_COUNT>(G, tab, mm, E, nodes, sh);
            bad |= o.bad;
            pts += o.start - 1 - (o.iexit > 0 ? o.iexit : 1);
            if (o.count < nodes) boe = E; else toe = E;
        }
        const double bottom = toe;
        // third bisection: the sign change of u(0) inside [bottom, top] (DFTAtom.cpp:513-534)
        double Top = top, Bot = bottom;
        int iter3 = 0, conv = 0;
        {
            const SweepOut o = scan_sweep<DFTA_SWEEP_ZERO>(G, tab, mm, Bot, 0, sh);
            ++n_zero; bad |= o.bad; pts += o.start - 2;
            const bool sgnBottom = o.u0 > 0;
            while (iter3 < kIter3) {
                const double E = (Top + Bot) / 2;
                const SweepOut z = scan_sweep<DFTA_SWEEP_ZERO>(G, tab, mm, E, 0, sh);
                ++n_zero; ++iter3; bad |= z.bad; pts += z.start - 2;
                const double Top_was = Top, Bot_was = Bot;
                if ((z.u0 > 0) == sgnBottom) Bot = E; else Top = E;
                const double ad = fabs(z.u0);
                if (Top - Bot < kErr && !(ad != ad) && ad < 1E15) { conv = 1; break; }
                if (fixed_point && Top == Top_was && Bot == Bot_was) {      // the same midpoint, sweep and decision to the cap (levels.hip)
                    const int rest = kIter3 - iter3;
                    n_zero += rest; n_fixed += rest; iter3 = kIter3;
                    break;
                }
            }
        }
        handed = Bot - 3;                                            // DFTAtom.cpp:541
        have_handed = true;
        if (threadIdx.x == 0) {
            J->top = top; J->bottom = bottom; J->toe = Top; J->boe = Bot; J->E = Bot;
            J->bottom0 = bottom0;
            J->converged = conv; J->n_count = n_count; J->n_zero = n_zero; J->iter3 = iter3; J->n_fixed = n_fixed;
            J->cur_len[1] = len2; J->n_points = pts; J->phase = kPhDone; J->haveSgn = 1;
            const int skipped = (nodes == 0 ? len2 : 0) + n_fixed;          // counted, not integrated
            atomicAdd(counters, (unsigned long long)(n_count + n_zero - skipped));
            atomicAdd(counters + 1, (unsigned long long)pts);
            if (bad) atomicOr(counters + 3, 1ull);
        }
    }
}

// interleaved tolerance-mode table of every slot: rows { V + c_l, 2 Rp^2 delta^2 e2 }
__global__ void k_scan_build_tab(double2* __restrict__ tabs, const double* __restrict__ V, const double* __restrict__ cl,
                                 const double* __restrict__ e2, const int* __restrict__ slot_v, const int* __restrict__ slot_l, int N,
                                 int logC, double twoRp2d2)
{
    const int slot = blockIdx.y;
    const int v = slot_v[slot], l = slot_l[slot];
    const size_t rows = (size_t)N;
    const int C = 1 << logC;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        double2 tt;
        tt.x = V[(size_t)v * N + i] + cl[(size_t)l * N + i];
        tt.y = twoRp2d2 * e2[i];
        const size_t pos = (i == N - 1) ? (size_t)(N - 1) : (((size_t)(i & (C - 1)) << 10) + (i >> logC));
        tabs[(size_t)slot * rows + pos] = tt;
    }
}

ScanGrid scan_grid_of(const dfta_grid* g)
{
    ScanGrid G;
    G.N = g->N;
    G.logC = g->levels - 10;
    G.delta = g->delta;
    G.far_thr = g->far_arg_threshold;
    G.c4 = g->delta2p4;
    G.r = g->d_r;
    return G;
}

}  // namespace

int dfta_scan_supported(const dfta_grid* g) { return g && !g->uniform && g->levels >= 12 && g->levels <= 24; }

int dfta_launch_scan_build_tab(dfta_ctx* ctx, const dfta_grid* g, double2* tabs, double2* mm, const double* dV, const int* d_slot_v, const int* d_slot_l, int nslots)
{
    hipLaunchKernelGGL(k_scan_build_tab, dim3(std::min(256, (g->N + 255) / 256), nslots), dim3(256), 0, ctx->stream, tabs, dV, g->d_cl, g->d_e2,
                       d_slot_v, d_slot_l, g->N, g->levels - 10, 2. * g->Rp2delta2);
    DFTA_CHECK_LAUNCH(ctx);
    hipLaunchKernelGGL(k_scan_lane_minmax, dim3(nslots), dim3(kT), 0, ctx->stream, tabs, g->N, g->levels - 10, mm);
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

int dfta_launch_scan_sweeps(dfta_ctx* ctx, const dfta_grid* g, int kind, int ntrials, const double2* tabs, const double2* mm, const int* d_trial_slot,
                            const double* dE, const int* dLimit, int* dCount, double* dU0, int* dStart, int* dTrip, int* dBad)
{
    hipLaunchKernelGGL(k_scan_sweeps, dim3(ntrials), dim3(kT), 0, ctx->stream, scan_grid_of(g), kind, tabs, mm, d_trial_slot, dE, dLimit, dCount, dU0,
                       dStart, dTrip, dBad);
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

int dfta_launch_scan_levels(dfta_ctx* ctx, const dfta_grid* g, dfta::Job* d_jobs, const int* d_chain_off, int nchains, int chained,
                            const double2* tabs, const double2* mm, int fixed_point, unsigned long long* d_counters)
{
    hipLaunchKernelGGL(k_scan_levels, dim3(nchains), dim3(kT), 0, ctx->stream, scan_grid_of(g), d_jobs, d_chain_off, chained, tabs, mm, fixed_point, d_counters);
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

// ---- C ABI: the tolerance-mode twin of dfta_numerov_sweeps -----------------------------------------------------------------
extern "C" int dfta_numerov_sweeps_scan(dfta_ctx* ctx, const dfta_grid* g, int kind, int nV, const double* V, int ntrials, const int* vidx,
                                        const int* l, const double* E, const int* nodesLimit, int* count_out, double* u0_out,
                                        int* start_out, int* trip_out, int* fallback_out)
{
    if (!ctx || !g) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, dfta_scan_supported(g), "the scan sweeps need a logarithmic grid of 12 .. 24 multigrid levels");
    DFTA_REQUIRE(ctx, V && l && E && nV > 0 && ntrials >= 0, "null input");
    DFTA_REQUIRE(ctx, kind == DFTA_SWEEP_COUNT || kind == DFTA_SWEEP_ZERO, "kind");
    DFTA_REQUIRE(ctx, kind != DFTA_SWEEP_COUNT || (nodesLimit && count_out), "COUNT needs nodesLimit and count_out");
    if (ntrials == 0) return DFTA_OK;
    const int N = g->N;
    std::vector<int> slot_v, slot_l, tslot(ntrials), lim(ntrials, 0);
    for (int q = 0; q < ntrials; ++q) {
        const int v = vidx ? vidx[q] : 0;
        DFTA_REQUIRE(ctx, v >= 0 && v < nV && l[q] >= 0 && l[q] <= 3, "vidx / l");
        int sl = -1;
        for (size_t k = 0; k < slot_v.size(); ++k) if (slot_v[k] == v && slot_l[k] == l[q]) sl = (int)k;
        if (sl < 0) { sl = (int)slot_v.size(); slot_v.push_back(v); slot_l.push_back(l[q]); }
        tslot[q] = sl;
        if (nodesLimit) { DFTA_REQUIRE(ctx, nodesLimit[q] >= 0 && nodesLimit[q] < (1 << 30), "nodesLimit out of range"); lim[q] = nodesLimit[q]; }
    }
    const int nslots = (int)slot_v.size();
    hipStream_t st = ctx->stream;
    DevBuf<double> dV, dE, dU0;
    DevBuf<int> dSv, dSl, dTs, dLim, dCount, dStart, dTrip, dBad;
    DevBuf<double2> dTab, dMm;
    DFTA_HIP(ctx, dV.alloc((size_t)nV * N));
    DFTA_HIP(ctx, dMm.alloc((size_t)nslots * kT));
    DFTA_HIP(ctx, hipMemcpyAsync(dV.p, V, (size_t)nV * N * sizeof(double), hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, dE.alloc(ntrials)); DFTA_HIP(ctx, dU0.alloc(ntrials));
    DFTA_HIP(ctx, dSv.alloc(nslots)); DFTA_HIP(ctx, dSl.alloc(nslots)); DFTA_HIP(ctx, dTs.alloc(ntrials)); DFTA_HIP(ctx, dLim.alloc(ntrials));
    DFTA_HIP(ctx, dCount.alloc(ntrials)); DFTA_HIP(ctx, dStart.alloc(ntrials)); DFTA_HIP(ctx, dTrip.alloc(ntrials)); DFTA_HIP(ctx, dBad.alloc(ntrials));
    DFTA_HIP(ctx, dTab.alloc((size_t)nslots * N));
    DFTA_HIP(ctx, hipMemcpyAsync(dE.p, E, sizeof(double) * ntrials, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipMemcpyAsync(dSv.p, slot_v.data(), sizeof(int) * nslots, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipMemcpyAsync(dSl.p, slot_l.data(), sizeof(int) * nslots, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipMemcpyAsync(dTs.p, tslot.data(), sizeof(int) * ntrials, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipMemcpyAsync(dLim.p, lim.data(), sizeof(int) * ntrials, hipMemcpyHostToDevice, st));
    int rc = dfta_launch_scan_build_tab(ctx, g, dTab.p, dMm.p, dV.p, dSv.p, dSl.p, nslots);
    if (rc) return rc;
    DFTA_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    rc = dfta_launch_scan_sweeps(ctx, g, kind, ntrials, dTab.p, dMm.p, dTs.p, dE.p, dLim.p, dCount.p, dU0.p, dStart.p, dTrip.p, dBad.p);
    if (rc) return rc;
    DFTA_HIP(ctx, hipEventRecord(ctx->ev[1], st));
    ctx->have_kernel_time = true;
    std::vector<int> hb(ntrials);
    if (count_out) DFTA_HIP(ctx, hipMemcpyAsync(count_out, dCount.p, sizeof(int) * ntrials, hipMemcpyDeviceToHost, st));
    if (u0_out) DFTA_HIP(ctx, hipMemcpyAsync(u0_out, dU0.p, sizeof(double) * ntrials, hipMemcpyDeviceToHost, st));
    if (start_out) DFTA_HIP(ctx, hipMemcpyAsync(start_out, dStart.p, sizeof(int) * ntrials, hipMemcpyDeviceToHost, st));
    if (trip_out) DFTA_HIP(ctx, hipMemcpyAsync(trip_out, dTrip.p, sizeof(int) * ntrials, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipMemcpyAsync(fallback_out ? fallback_out : hb.data(), dBad.p, sizeof(int) * ntrials, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    return DFTA_OK;
}
