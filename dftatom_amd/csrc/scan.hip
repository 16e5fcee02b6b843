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
// trial is spread over the 512 lanes of a workgroup: lane t multiplies the matrices of its own C = (N-1)/512 grid points (two
// independent columns: 2 fma + 2 add per point, g from the table row in 7 instructions), a log-depth scan over the lanes combines
// the 512 segment matrices (wave shuffles + one hand-over through LDS), and -- CountNodes only -- a second pass over the rows of
// the classically allowed lanes with the now known incoming (w, D) counts the sign changes.  A 131 073-point sweep takes 35 (u(0)) ...
// 75 us (count) on one compute unit instead of 4 ms as a dependent chain, so the bisections of a level need no speculation: ONE
// workgroup runs LocateInterval and the u(0) bisection of its level from start to end on the device (k_scan_levels), ~150 sweeps back
// to back, no host round trips.  What bounds it: fp64 VALU issue of the one compute unit (~20 instructions per row and lane).
// The summed form carries the slope D as a variable of its own (the reference forms it as 2w - w', a difference of numbers that agree
// to 3-4 digits): tests/test_scan_precision.py shows it ~1000x closer to the 80-bit eigenvalue than the reference's own double
// arithmetic -- the gate against the exact kernels (6e-11 |E| + 6e-10 Ha) measures the reference's rounding bias.
//
// Table: per slot (potential, l) the rows veff_i (8 bytes), f_i = A_i (veff_i - E) + delta^2/4 with A_i = 2 Rp^2 delta^2 exp(2 i delta)
// (Numerov.h:96-101) factored per grid as Atop[t] T[k]; LANE-INTERLEAVED: row i = t C + k is stored at k 512 + t, so that the 512 lanes
// read consecutive addresses at every step (the layout idea of the multigrid levels).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "internal.h"
#include "levels.h"

#ifdef DFTA_SCAN_PROF
// -DDFTA_SCAN_PROF: wave 0 accumulates the clock ticks (100 MHz wall clock) spent in the sections of scan_sweep; dfta_debug_scan_prof reads them
__device__ unsigned long long g_scan_prof[16];
#define SCAN_TICK(slot) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); atomicAdd(&g_scan_prof[slot], now_ - tick_); tick_ = now_; } } while (0)
#define SCAN_TICK0() unsigned long long tick_ = wall_clock64()
#else
#define SCAN_TICK(slot) do { } while (0)
#define SCAN_TICK0() do { } while (0)
#endif

namespace {

constexpr int kLogT = 9;
constexpr int kT = 1 << kLogT;    // lanes per trial = threads per workgroup: 8 waves, two per SIMD, 256 VGPRs each
constexpr int kW = kT / 64;       // waves per workgroup
constexpr double kInv12 = 1. / 12.;
constexpr int kScanPadRows = 32;  // rows of padding in front of the veff tables (dfta_scan_tables::tabv_alloc)

// Tables (device): veff rows lane-interleaved per slot (N doubles: row i = t C + k at k 512 + t, row N-1 at N-1), per slot and lane
// {min, max} of veff, and per GRID the factors of A_i = 2 Rp^2 delta^2 exp(2 i delta) = Atop[t] T[k]: Atop[t] = A of the lane's top
// row (Atop[512] = A_{N-1}), T[k] = exp(-2 delta (C-1-k)) -- so that a row costs 8 bytes of L2 traffic, not 16.
struct ScanGrid {
    int N, logC;                  // N - 1 = kT << logC
    double delta, far_thr, c4;    // delta, exp(a) < 1e-200 <=> a < far_thr, delta^2/4
    double Rp;
    const double* r;              // r_i
    const double* Atop;           // 1025
    const double* T;              // C
};

struct ScanMatchArgs {            // the match solve at the end of k_scan_levels
    int mode;                     // 0: none (the exact k_match follows), 1: Psi as Numerov.h:403-504 returns it, 2: normalised too (Simpson 3/8)
    double* Psi;                  // njobs * N
    const double *eh, *cnst;      // exp(i delta/2), Rp delta exp(delta i)   (DFTAtom.cpp:42,47)
    double zero1[4];              // GetBoundaryValueZero(1, l)
    int* jstart_keep;             // per job: cut-off index of the matched solve (-1: frozen)
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

// g = f / (1 - f/12) = f (1 + x)(1 + x^2)(1 + x^4) + O(x^8), x = f/12.  Which form a lane uses is decided BEFORE its rows are read, from
// a bound on |x| over its rows (A <= Atop, veff in [min, max]): V4 for |x| <= 2^-14.5 (x^4 < 2^-58), V8 for |x| <= 2^-7, else an IEEE division.
enum GVar { V4 = 0, V8 = 1, VDIV = 2 };
template <int V>
__device__ __forceinline__ double g_of(double f)
{
    const double x = f * kInv12;
    if (V == VDIV) return f / (1. - x);
    const double x2 = x * x;
    const double t1 = fma(f, x, f);
    const double t2 = fma(t1, x2, t1);
    if (V == V4) return t2;
    return fma(t2, x2 * x2, t2);
}
constexpr double kX4 = 4.3e-5, kX8 = 0.0078125;

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
    Mat wave_tot[2][kW];          // double-buffered by sweep parity: a wave is never two sweeps ahead (every sweep has a barrier)
    int wave_bad[2][kW];
    Turn wave_turn[2][kW];
    int red[2][kW];
    unsigned long long xres[16];  // k_scan_levels_group: the round's results of the job's members
};

struct LaneState {     // what scan_sweep leaves in every lane (the match solve goes on from there)
    double w_in, D_in;  // state (w_i, w_i - w_{i+1}) at the lane's top step row i = ibase + khi
    int khi, klo;       // the lane's step rows [klo, khi] (empty: khi < klo)
    int var;            // form of g for the wave's rows
    double us, us1, w_fin, D_fin;
};

struct SweepOut {
    int count;          // COUNT: min(sign changes, limit + 1) (+ the final extrapolated test); the decision value of CountNodes
    int start, iexit;   // cut-off index, index at which CountNodes returned (0: ran to the end)
    double u0;          // ZERO: u_1 (2 + f_1) - u_2 (Numerov.h:398)
    int bad;            // a non-finite value or f >= 12 in a step row: the exact kernels must decide this trial
};

// i is the same in every lane: the table value comes through the scalar cache
__device__ __forceinline__ double far_arg_s(const double* __restrict__ r, int i, double s, double delta)
{
    const int iu = __builtin_amdgcn_readfirstlane(i);
    return -r[iu] * s - static_cast<double>(iu) * delta * 0.5;      // Numerov.h:107
}

// GetMaxRadiusIndex (Numerov.h:119-136): the start value exp(-r_i sqrt(2|E|) - i delta/2) falls monotonically with i, so the integer
// bisection returns the smallest index >= 2 whose start value is below 1e-200 (N-1 if none).  Settled on the table values themselves,
// starting from the cut-off of the previous trial of the bisection (it moves by less than a cell after the first few halvings) or from
// the root of Rp sq (e^x - 1) + x/2 = -far_thr, x = i delta (Newton's method).  Every lane does the same.
__device__ int scan_cutoff(const ScanGrid& G, double sq, int hint)
{
    auto below = [&](int i) { return far_arg_s(G.r, i, sq, G.delta) < G.far_thr; };
    auto settle = [&](int i0, int maxsteps, bool& ok) {
        int n = 0;
        const bool b1 = i0 > 2 && below(i0 - 1), b0 = i0 >= G.N - 1 || below(i0);     // both table values in flight at once
        if (!b1 && b0) { ok = true; return i0; }
        while (n < maxsteps && i0 > 2 && below(i0 - 1)) { --i0; ++n; }
        while (n < maxsteps && i0 < G.N - 1 && !below(i0)) { ++i0; ++n; }
        ok = n < maxsteps;
        return i0;
    };
    bool ok = false;
    int i0 = 0;
    if (hint >= 2 && hint <= G.N - 1) i0 = settle(hint, 3, ok);
    if (ok) return i0;
    const double c = -G.far_thr, a = G.Rp * sq;
    i0 = G.N - 1;
    if (a > 0.) {
        double x = log(c / a + 1.);
        for (int it = 0; it < 4; ++it) {
            const double ex = exp(x);
            x -= (a * (ex - 1.) + 0.5 * x - c) / (a * ex + 0.5);
        }
        const double fi = ceil(x / G.delta);
        i0 = fi < 2. ? 2 : (fi > static_cast<double>(G.N - 1) ? G.N - 1 : static_cast<int>(fi));
    }
    return settle(i0, 1 << 30, ok);
}

// ---- the row loops --------------------------------------------------------------------------------------------------------------
// pv: the lane's column of the slot's veff table (row k at pv[k << kLogT]); A_k = Atop T[k]; f = A (veff - E) + c4.
// gam = 1: the row is taken; gam = 0: the state stays as it is (rows of a lane outside [lo, s-1]; exact either way: fma(1, x, y) = x + y)
struct Pass1 {        // transfer matrix of the rows, two columns
    Mat M;
    template <int V> __device__ __forceinline__ void row(double f)
    {
        const double g = g_of<V>(f);
        M.c = fma(g, M.a, M.c); M.a += M.c;
        M.d = fma(g, M.b, M.d); M.b += M.d;
    }
    template <int V> __device__ __forceinline__ void row_gated(double f, double gam)
    {
        const double g = gam * g_of<V>(f);
        M.c = fma(g, M.a, M.c); M.a = fma(gam, M.c, M.a);
        M.d = fma(g, M.b, M.d); M.b = fma(gam, M.d, M.b);
    }
};
struct Pass2 {        // sign changes of w along the rows, from (w, D); per lane
    double w, D;
    int cnt;
    template <int V> __device__ __forceinline__ void row(double f)
    {
        const double g = g_of<V>(f);
        D = fma(g, w, D);
        const double wn = w + D;
        cnt += ((wn > 0.) != (w > 0.));
        w = wn;
    }
    template <int V> __device__ __forceinline__ void row_gated(double f, double gam)
    {
        const double g = gam * g_of<V>(f);
        D = fma(g, w, D);
        const double wn = fma(gam, D, w);
        cnt += ((wn > 0.) != (w > 0.));
        w = wn;
    }
};
struct Pass2W {       // the same for a wave whose lanes all run all their rows: the changes are counted per wave in scalar registers
    double w, D;
    unsigned long long prev;
    int cnt;
    template <int V> __device__ __forceinline__ void row(double f)
    {
        const double g = g_of<V>(f);
        D = fma(g, w, D);
        w = w + D;
        const unsigned long long m = __ballot(w > 0.);
        cnt += __popcll(m ^ prev);
        prev = m;
    }
    template <int V> __device__ __forceinline__ void row_gated(double f, double) { row<V>(f); }
};

// rows [klo, khi] of one lane (per-lane bounds), descending, eight rows in flight: small grids (C < 16) and the innermost rows.
// tabv: the slot's table (wave-uniform pointer), t: the lane's column.
template <int V, typename P>
__device__ __forceinline__ void rows_lane(const double* __restrict__ tabv, int t, const double* __restrict__ T, double Atop, int khi, int klo, double E, double c4, P& ps)
{
    int k = khi;
    for (; k - 7 >= klo; k -= 8) {
        double v[8], tk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[j] = tabv[((size_t)(k - j) << kLogT) + t]; tk[j] = T[k - j]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) ps.template row<V>(fma(Atop * tk[j], v[j] - E, c4));
    }
    if (k >= klo) {
        double v[8], tk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int kk = k - j >= klo ? k - j : klo; v[j] = tabv[((size_t)kk << kLogT) + t]; tk[j] = T[kk]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (k - j >= klo) ps.template row<V>(fma(Atop * tk[j], v[j] - E, c4));
    }
}

// The lane's column of a slot's table as a BUFFER: address = descriptor base (scalar) + the row's byte offset (ONE scalar register) + the
// lane's byte offset (a vector register that never changes) -- `buffer_load_dwordx2 v, voff, rsrc, soff offen`: no vector address
// arithmetic per row (a global load needs a 64-bit per-lane address: two VALU instructions per row, which this loop cannot afford).
// The descriptor starts kScanPadRows rows in front of the slot's row 0, so that rows -kScanPadRows .. C-1 have offsets >= 0.
struct RowBuf {
#if defined(__HIP_DEVICE_COMPILE__)
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff;
    __device__ __forceinline__ RowBuf(const double* tabv, int t, int C)
    {
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(tabv) - (size_t)kScanPadRows * kT, 0,
                                               static_cast<int>(sizeof(double) * ((size_t)(C + kScanPadRows) * kT + 1)), 0x00020000);
        voff = static_cast<unsigned>(t) * 8u;
    }
    __device__ __forceinline__ double row(int k) const      // k wave-uniform
    {
        typedef unsigned int u2 __attribute__((ext_vector_type(2)));
        const u2 r = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, (k + kScanPadRows) * (kT * 8), 0);
        return __builtin_bit_cast(double, r);
    }
#else       // host pass: parsed, never run
    const double* p;
    __device__ __forceinline__ RowBuf(const double* tabv, int t, int) : p(tabv + t) {}
    __device__ __forceinline__ double row(int k) const { return p[(size_t)k * kT]; }
#endif
};

// all C rows of the wave's lanes (C a multiple of 8 NB), descending, NB batches of eight rows in flight (the loads of a batch are issued
// NB - 1 batches ahead of its arithmetic: a wave keeps 4 NB KB on the way -- what bounds a row loop is latency x bytes in flight, not issue);
// the row pointer and T[k] are wave-uniform (scalar address arithmetic, scalar loads).  PRED: a lane takes only its rows in [klo, khi] (the waves that hold the
// cut-off, the exit point or the innermost lane) -- by a 0/1 factor, not by branches.
template <int V, bool PRED, int NB, typename P>
__device__ __forceinline__ void rows_wave(const double* __restrict__ tabv, int t, const double* __restrict__ T, double Atop, int C, int khi, int klo, double E,
                                          double c4, P& ps)
{
    static_assert(NB % 2 == 0 && 8 * (NB - 1) <= kScanPadRows, "two sets of T registers in turn; the loads ahead of row 0 stay inside the padding");
    typedef double d8 __attribute__((ext_vector_type(8)));
    // What bounds this loop is the wave's ISSUE rate (one instruction of any kind per 4 cycles): address arithmetic is kept out of it.
    // Per row: one scalar add (the row's byte offset) and one buffer load (RowBuf); per batch of eight rows ONE scalar load of the eight
    // factors T.  Nothing is clamped: the tables are padded in front.
    const RowBuf rb(tabv, t, C);
    auto tvec = [&](int k0) { return *reinterpret_cast<const d8*>(T + (k0 - 7)); };      // T[k0 - 7 .. k0], 64-byte aligned, ONE scalar load
    double v[NB][8];
    d8 tk[2];
#pragma unroll
    for (int b = 0; b < NB - 1; ++b)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[b][j] = rb.row(C - 1 - 8 * b - j);
    tk[0] = tvec(C - 1);
    for (int k = C - 1; k >= 0; k -= 8 * NB) {
        const int ku = __builtin_amdgcn_readfirstlane(k);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            constexpr int kAhead = NB - 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[(b + kAhead) % NB][j] = rb.row(ku - 8 * (b + kAhead) - j);     // below row 0 (last turn): the padding, unused
            tk[(b + 1) & 1] = tvec(ku - 8 * (b + 1));                                                     // the factors of the NEXT batch
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kr = ku - 8 * b - j;
                const double f = fma(Atop * tk[b & 1][7 - j], v[b][j] - E, c4);
                if (PRED) ps.template row_gated<V>(f, (kr <= khi && kr >= klo) ? 1. : 0.); else ps.template row<V>(f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// the innermost rows (index < 16) are always divided: there f -> l(l+1)/i^2 is far outside the series' range (row 1 of l = 3 has f > 12)
constexpr int kInnerRows = 16;
#ifndef DFTA_SCAN_ROW_BATCHES
#define DFTA_SCAN_ROW_BATCHES 4
#endif
constexpr int kRowBatches = DFTA_SCAN_ROW_BATCHES;     // eight-row batches in flight per wave on grids with C >= 8 kRowBatches rows per lane

template <int V, typename P>
__device__ __forceinline__ void run_rows_v(const double* __restrict__ tabv, int t, const double* __restrict__ T, double Atop, int C, int khi, int klo, bool full,
                                           bool inner, double E, double c4, P& ps)
{
    if (C >= 16) {
        const int klo_w = inner ? max(klo, kInnerRows) : klo;
        if (C >= 8 * kRowBatches) {
            if (full) rows_wave<V, false, kRowBatches>(tabv, t, T, Atop, C, khi, klo, E, c4, ps);
            else rows_wave<V, true, kRowBatches>(tabv, t, T, Atop, C, khi, klo_w, E, c4, ps);
        } else {
            if (full) rows_wave<V, false, 2>(tabv, t, T, Atop, C, khi, klo, E, c4, ps);
            else rows_wave<V, true, 2>(tabv, t, T, Atop, C, khi, klo_w, E, c4, ps);
        }
        if (inner && klo < kInnerRows && khi >= klo) rows_lane<VDIV>(tabv, t, T, Atop, min(khi, kInnerRows - 1), klo, E, c4, ps);
        return;
    }
    if (khi < klo) return;
    if (inner) rows_lane<VDIV>(tabv, t, T, Atop, khi, klo, E, c4, ps);
    else rows_lane<V>(tabv, t, T, Atop, khi, klo, E, c4, ps);
}

// `full` and `var` are wave-uniform
template <typename P>
__device__ __forceinline__ void run_rows(const double* __restrict__ tabv, int t, const double* __restrict__ T, double Atop, int C, int khi, int klo, bool full,
                                         bool inner, int var, double E, double c4, P& ps)
{
    if (var == V4) run_rows_v<V4>(tabv, t, T, Atop, C, khi, klo, full, inner, E, c4, ps);
    else if (var == V8) run_rows_v<V8>(tabv, t, T, Atop, C, khi, klo, full, inner, E, c4, ps);
    else run_rows_v<VDIV>(tabv, t, T, Atop, C, khi, klo, full, inner, E, c4, ps);
}

// One sweep of one trial by the whole workgroup.  KIND: DFTA_SWEEP_COUNT / DFTA_SWEEP_ZERO.  tabv: the slot's interleaved veff table;
// mm: per lane {min, max} of veff over the lane's rows ({-inf, +inf} when a row is NaN).
template <int KIND>
__device__ __noinline__ SweepOut scan_sweep(const ScanGrid& G_, const double* __restrict__ tabv_, const double2* __restrict__ mm_, double E, int limit, ScanShared& sh_, unsigned par,
                               int hint = 0, LaneState* ls = nullptr)
{
    // this is a real function (two instances, called ~150 times per level): its pointers arrive without an address space (common.h)
    ScanGrid G = G_;
    G.r = dfta_as_constant(dfta_uniform(G_.r)); G.Atop = dfta_as_constant(dfta_uniform(G_.Atop)); G.T = dfta_as_constant(dfta_uniform(G_.T));
    G.N = __builtin_amdgcn_readfirstlane(G_.N); G.logC = __builtin_amdgcn_readfirstlane(G_.logC);      // the same in every lane: say so (scalar registers)
    const double* __restrict__ tabv = dfta_as_constant(dfta_uniform(tabv_));
    const double2* __restrict__ mm = dfta_as_constant(dfta_uniform(mm_));
    ScanShared& sh = *dfta_as_shared(&sh_);
    const int tid = threadIdx.x;
    const int t = kT - 1 - tid;                 // segment of this lane: thread order = sweep order (descending index)
    const int C = 1 << G.logC;
    const int lane = tid & 63, wv = tid >> 6;
    SweepOut out;
    out.count = 0; out.u0 = 0; out.bad = 0; out.iexit = 0;
    SCAN_TICK0();
    const double sq = sqrt(2. * fabs(E));
    const int s = scan_cutoff(G, sq, hint);
    SCAN_TICK(0);
    out.start = s;
    const double us = exp(far_arg_s(G.r, s, sq, G.delta));          // GetBoundaryValueFar at the cut-off and one point inside
    const double us1 = exp(far_arg_s(G.r, s - 1, sq, G.delta));
    auto f_of_row = [&](int ii) -> double {          // ii is the same in every lane: scalar loads
        const int i = __builtin_amdgcn_readfirstlane(ii);
        const bool last = (i == (kT << G.logC));
        const double v = last ? tabv[(size_t)kT << G.logC] : tabv[((size_t)(i & (C - 1)) << kLogT) + (i >> G.logC)];
        const double A = last ? G.Atop[kT] : G.Atop[i >> G.logC] * G.T[i & (C - 1)];
        return fma(A, v - E, G.c4);
    };
    const double f1 = f_of_row(1), f2 = f_of_row(2);          // for the end of the sweep: in flight from here
    const double fs = f_of_row(s), fs1 = f_of_row(s - 1);
    const double ws = (1. - kInv12 * fs) * us, ws1 = (1. - kInv12 * fs1) * us1;        // Numerov.h:297,302
    const int ibase = t << G.logC;
    const double Atop = G.Atop[t];
    const double2 m = mm[t];
    SCAN_TICK(1);
    // ---- CountNodes: where does the sweep leave through the classical turning point (Numerov.h:337-341)?  The loop index runs over
    // [1, s-2]; lanes whose rows lie on one side of E answer from {min, max}; the rows of the others are looked at by their whole wave
    int iexit = 0;
    if (KIND == DFTA_SWEEP_COUNT) {
        Turn tu = {-1, -1, -1};
        const int top = min(s - 2, ibase + C - 1), bot = max(1, ibase);
        int cls = 0;
        if (top >= bot) {
            if (m.x > E) { tu.fmax = top; cls = 1; }         // every row forbidden
            else if (m.y <= E) { tu.amax = top; cls = 2; }   // every row allowed
            else cls = 3;
        }
        unsigned long long mixed = __ballot(cls == 3);
        while (mixed) {
            const int src = __ffsll(static_cast<long long>(mixed)) - 1;
            mixed &= mixed - 1;
            const int ts = kT - 1 - ((tid & ~63) + src), ib = ts << G.logC;
            const int tp = __shfl(top, src, 64), bt = __shfl(bot, src, 64);
            Turn acc = {-1, -1, -1};
            for (int base = (tp - ib) & ~63; base >= 0 && ib + base + 63 >= bt; base -= 64) {
                const int k = base + lane, i = ib + k;
                const bool valid = k < C && i <= tp && i >= bt;
                const double tt = valid ? tabv[((size_t)k << kLogT) + ts] - E : 0.;
                const unsigned long long Am = __ballot(valid && tt <= 0.), Fm = __ballot(valid && tt > 0.);
                Turn ch = {-1, -1, -1};
                if (Am) {
                    const int ba = 63 - __clzll(static_cast<long long>(Am));
                    ch.amax = ib + base + ba;
                    const unsigned long long below = ba ? (Fm & ((1ull << ba) - 1ull)) : 0ull;
                    if (below) ch.fbelow = ib + base + 63 - __clzll(static_cast<long long>(below));
                }
                if (Fm) ch.fmax = ib + base + 63 - __clzll(static_cast<long long>(Fm));
                acc = turn_combine(acc, ch);
            }
            if (lane == src) tu = acc;
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {     // inclusive scan of the summaries: lane 63 ends up with the wave's
            Turn X;
            X.amax = __shfl_up(tu.amax, off, 64); X.fmax = __shfl_up(tu.fmax, off, 64); X.fbelow = __shfl_up(tu.fbelow, off, 64);
            if (lane >= off) tu = turn_combine(X, tu);
        }
        if (lane == 63) sh.wave_turn[par][wv] = tu;
        __syncthreads();
        Turn all = sh.wave_turn[par][0];
#pragma unroll
        for (int q = 1; q < kW; ++q) all = turn_combine(all, sh.wave_turn[par][q]);
        iexit = all.fbelow >= 0 ? all.fbelow : 0;       // the first forbidden index below the first allowed one; 0: none
    }
    out.iexit = iexit;
    SCAN_TICK(2);
    // ---- pass 1: the segment's transfer matrix (two columns) over its step rows in [lo, s-1]; step i maps (w_i, D_i) to (w_{i-1}, D_{i-1})
    const int lo = max(iexit + 1, 2), hi = s - 1;
    const int khi = min(hi - ibase, C - 1), klo = max(lo - ibase, 0);
    const bool inner = (t == 0);                     // the lane that owns the innermost rows
    const bool covered = (khi == C - 1) && (klo == 0) && !inner;
    const bool full = __all(covered) != 0;
    // which form of g: from the bound |f| <= Atop max(|min - E|, |max - E|) + c4 over the lane's rows
    const double xb = (Atop * fmax(fabs(m.x - E), fabs(m.y - E)) + G.c4) * kInv12;
    const int myvar = xb <= kX4 ? V4 : (xb <= kX8 ? V8 : VDIV);
    const bool has_rows = khi >= klo;
    int var = V4;
    if (__any(has_rows && !inner && myvar != V4)) var = __any(has_rows && !inner && myvar == VDIV) ? VDIV : V8;
    // f >= 12 in a step row (d <= 0: u and w differ in sign) is left to the exact kernels; the innermost lane looks at its rows itself
    int bad = has_rows && !inner && !(xb < 1.);
    Pass1 p1;
    p1.M.a = 1.; p1.M.b = 0.; p1.M.c = 0.; p1.M.d = 1.;
    if (__any(has_rows)) run_rows(tabv, t, G.T, Atop, C, khi, klo, full, inner, var, E, G.c4, p1);
    const Mat M = p1.M;
    SCAN_TICK(3);
    // ---- combine: inclusive scan of the matrices in thread order (wave shuffles, then the 16 wave totals through LDS)
    Mat P = M;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        Mat Ee;
        Ee.a = shfl_up_d(P.a, off); Ee.b = shfl_up_d(P.b, off); Ee.c = shfl_up_d(P.c, off); Ee.d = shfl_up_d(P.d, off);
        if (lane >= off) P = mat_mul(P, Ee);
    }
    const int wbad = __any(bad) != 0;
    if (lane == 63) { sh.wave_tot[par][wv] = P; sh.wave_bad[par][wv] = wbad; }
    SCAN_TICK(4);
    __syncthreads();
    SCAN_TICK(5);
    // incoming state of this wave, and the final state of the sweep
    double w_in = ws1, D_in = ws1 - ws;
    double w_fin, D_fin;
    int anybad = 0;
    {
        double w = ws1, D = ws1 - ws;
#pragma unroll
        for (int q = 0; q < kW; ++q) {
            if (q == wv) { w_in = w; D_in = D; }
            const Mat Wq = sh.wave_tot[par][q];
            anybad |= sh.wave_bad[par][q];
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
    anybad |= !(fabs(w_fin) < INFINITY) || !(fabs(D_fin) < INFINITY);
    // u_1, u_2 and the extrapolation to the origin (Numerov.h:343-346,398) when the sweep ran to index 1: the final state is
    // (w_1, w_1 - w_2).  Row 1 of l = 3 has f > 12 (d < 0): these last values are divided exactly
    double u1 = 0, u2 = 0;
    if (iexit <= 1) {
        anybad |= !(f2 < 12.);
        u1 = w_fin / (1. - kInv12 * f1); u2 = (w_fin - D_fin) / (1. - kInv12 * f2);
        out.u0 = u1 * (2. + f1) - u2;
    }
    SCAN_TICK(6);
    if (KIND == DFTA_SWEEP_COUNT) {
        // ---- pass 2: sign changes between w_i and w_{i-1} for the step rows i in [lo2, s-1] (u has the sign of w there: d > 0; the
        // comparison of step 2, u_2 against u_1, is taken from the divided values above).  A lane whose rows are all forbidden (g >= 0:
        // w'' = g w keeps |w| convex) crosses zero at most once: its count is the sign change between its two ends, known from M.
        const int lo2 = max(lo, 3);
        const int klo2 = max(lo2 - ibase, 0);
        int cnt = 0;
        const bool rows2 = khi >= klo2;
        const bool convex = rows2 && (m.x >= E) && klo2 == klo && !inner;     // veff >= E on every row: f >= c4 > 0
        if (convex) {
            const double w_out = fma(M.a, w_in, M.b * D_in);
            cnt = (w_out > 0.) != (w_in > 0.);
        }
        const bool need = rows2 && !convex;
        if (__any(need)) {
            if (full && C >= 16 && klo2 == klo) {          // every lane of the wave runs all its rows: per-wave count (the convex lanes' rows included)
                Pass2W p2;
                p2.w = w_in; p2.D = D_in; p2.cnt = 0; p2.prev = __ballot(w_in > 0.);
                run_rows(tabv, t, G.T, Atop, C, khi, klo2, true, false, var, E, G.c4, p2);
                cnt = lane == 0 ? p2.cnt : 0;
            } else {                                       // per-lane counts; the convex lanes keep the count of their two ends
                Pass2 p2;
                p2.w = w_in; p2.D = D_in; p2.cnt = 0;
                run_rows(tabv, t, G.T, Atop, C, need ? khi : -1, klo2, false, inner, var, E, G.c4, p2);
                if (need) cnt = p2.cnt;
            }
        }
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) sh.red[par][wv] = cnt;
        SCAN_TICK(7);
        __syncthreads();
        SCAN_TICK(8);
        int total = 0;
#pragma unroll
        for (int q = 0; q < kW; ++q) total += sh.red[par][q];
        if (iexit <= 1 && s - 1 >= 2) total += ((u1 > 0.) != (u2 > 0.));       // step 2
        // the first comparison of the loop is against the sign of the start value u_{s-1} = us1 > 0 -- that of w_{s-1}, the incoming w
        // of the first lane.  Early return at count > limit: the value is limit + 1.
        if (total > limit) total = limit + 1;
        else if (iexit == 0) total += ((out.u0 > 0.) != (u1 > 0.));      // ran to the end: the extrapolated point (Numerov.h:343-347)
        out.count = total;
    }
    out.bad = anybad;
    if (ls) { ls->w_in = w_in; ls->D_in = D_in; ls->khi = khi; ls->klo = klo; ls->var = var; ls->us = us; ls->us1 = us1; ls->w_fin = w_fin; ls->D_fin = D_fin; }
    return out;
}

// ---- SolveSchrodingerMatchSolutionCompletely (Numerov.h:403-504) + NormalizeNonUniform (DFTAtom.cpp:36-56) by the workgroup --------------
// Inward from the cut-off to the first local maximum of u (the match point), outward from the origin up to it, the outer part rescaled so
// that the two meet -- both integrations as scans.  Psi (N doubles, plain layout) receives the reference's Psi; with fused_norm the
// wave function comes out normalised (Simpson 3/8 of Psi^2 Rp delta e^{delta i} as one parallel sum; the other rules stay with
// k_normalize).  Returns the match point (< 2: the scan hands the level back to the exact kernels).
struct MatchOut { int matchPoint, start, bad; };

// u = w / (1 - x), x = f/12, by the series of the row's form of g
template <int V>
__device__ __forceinline__ double u_of(double w, double f)
{
    const double x = f * kInv12;
    if (V == VDIV) return w / (1. - x);
    const double x2 = x * x;
    const double t1 = fma(w, x, w);
    const double t2 = fma(t1, x2, t1);
    if (V == V4) return t2;
    return fma(t2, x2 * x2, t2);
}

template <int V>
__device__ __forceinline__ void match_in_rows(const double* __restrict__ tabv, int t, const double* __restrict__ T, double Atop, int ibase, int khi, int klo,
                                              double E, double c4, double w, double D, double uprev, int top_checked, double* __restrict__ Psi, int& cand)
{
    // rows khi .. klo, descending: at row i the state is (w_i, D_i): u_i = w_i / d_i is stored and compared with u_{i+1}
    int k = khi;
    while (k >= klo) {
        double v[8], tk[8];
        const int nb = min(8, k - klo + 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int kk = j < nb ? k - j : klo; v[j] = tabv[((size_t)kk << kLogT) + t]; tk[j] = T[kk]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < nb) {
                const int i = ibase + k - j;
                const double f = fma(Atop * tk[j], v[j] - E, c4);
                const double u = u_of<V>(w, f);
                Psi[i] = u;
                if (i <= top_checked && cand < 0 && (u < uprev || fabs(u) > 1E15)) cand = i;      // Numerov.h:455
                uprev = u;
                const double g = g_of<V>(f);
                D = fma(g, w, D);
                w = w + D;
            }
        }
        k -= nb;
    }
}

template <int V, bool WRITE>
__device__ __forceinline__ void match_out_rows(const double* __restrict__ tabv, int t, const double* __restrict__ T, double Atop, int ibase, int klo, int khi,
                                               double E, double c4, Mat& M, double& w, double& D, double* __restrict__ Psi)
{
    // rows klo .. khi, ASCENDING: step i maps (w_i, w_i - w_{i-1}) to (w_{i+1}, w_{i+1} - w_i); WRITE: u_i is stored (rows >= 2), else the matrix
    int k = klo;
    while (k <= khi) {
        double v[8], tk[8];
        const int nb = min(8, khi - k + 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int kk = j < nb ? k + j : khi; v[j] = tabv[((size_t)kk << kLogT) + t]; tk[j] = T[kk]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < nb) {
                const int i = ibase + k + j;
                const double f = fma(Atop * tk[j], v[j] - E, c4);
                const double g = g_of<V>(f);
                if (WRITE) {
                    if (i >= 2) Psi[i] = u_of<V>(w, f);
                    D = fma(g, w, D);
                    w = w + D;
                } else {
                    M.c = fma(g, M.a, M.c); M.a += M.c;
                    M.d = fma(g, M.b, M.d); M.b += M.d;
                }
            }
        }
        k += nb;
    }
}

__device__ __noinline__ MatchOut scan_match(const ScanGrid& G_, const double* __restrict__ tabv_, const double2* __restrict__ mm, double E, double zero1, ScanShared& sh_,
                               unsigned& par, int hint, double* __restrict__ Psi_, const double* __restrict__ eh_, const double* __restrict__ cnst_, int fused_norm)
{
    ScanGrid G = G_;
    G.r = dfta_as_constant(dfta_uniform(G_.r)); G.Atop = dfta_as_constant(dfta_uniform(G_.Atop)); G.T = dfta_as_constant(dfta_uniform(G_.T));
    G.N = __builtin_amdgcn_readfirstlane(G_.N); G.logC = __builtin_amdgcn_readfirstlane(G_.logC);      // the same in every lane: say so (scalar registers)
    const double* __restrict__ tabv = dfta_as_constant(dfta_uniform(tabv_));
    double* __restrict__ Psi = dfta_as_global(dfta_uniform(Psi_));
    const double* __restrict__ eh = dfta_as_constant(dfta_uniform(eh_));
    const double* __restrict__ cnst = dfta_as_constant(dfta_uniform(cnst_));
    ScanShared& sh = *dfta_as_shared(&sh_);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int C = 1 << G.logC, N = G.N;
    MatchOut mo;
    // ---- inward: the sweep of SolutionInZero leaves every lane's incoming state
    LaneState ls;
    const SweepOut so = scan_sweep<DFTA_SWEEP_ZERO>(G, tabv, mm, E, 0, sh, par++ & 1, hint, &ls);
    const int s = so.start;
    mo.start = s; mo.bad = so.bad;
    auto f_of_row = [&](int ii) -> double {
        const int i = __builtin_amdgcn_readfirstlane(ii);
        const bool last = (i == (kT << G.logC));
        const double v = last ? tabv[(size_t)kT << G.logC] : tabv[((size_t)(i & (C - 1)) << kLogT) + (i >> G.logC)];
        const double A = last ? G.Atop[kT] : G.Atop[i >> G.logC] * G.T[i & (C - 1)];
        return fma(A, v - E, G.c4);
    };
    {
        const int t = kT - 1 - tid, ibase = t << G.logC;
        int cand = -1;
        if (ls.khi >= ls.klo) {
            // u of the row above the lane's top row: from the incoming state, w_{i+1} = w_i - D_i
            const int itop = ibase + ls.khi;
            const double fa = (itop + 1 <= N - 1) ? fma((itop + 1 == N - 1 ? G.Atop[kT] : G.Atop[(itop + 1) >> G.logC] * G.T[(itop + 1) & (C - 1)]),
                                                        (itop + 1 == N - 1 ? tabv[(size_t)kT << G.logC] : tabv[((size_t)((itop + 1) & (C - 1)) << kLogT) + ((itop + 1) >> G.logC)]) - E, G.c4) : 0.;
            const double uabove = (ls.w_in - ls.D_in) / (1. - kInv12 * fa);
            const bool innerl = (t == 0);
            const int var = innerl ? VDIV : ls.var;
            if (var == V4) match_in_rows<V4>(tabv, t, G.T, G.Atop[t], ibase, ls.khi, ls.klo, E, G.c4, ls.w_in, ls.D_in, uabove, s - 2, Psi, cand);
            else if (var == V8) match_in_rows<V8>(tabv, t, G.T, G.Atop[t], ibase, ls.khi, ls.klo, E, G.c4, ls.w_in, ls.D_in, uabove, s - 2, Psi, cand);
            else match_in_rows<VDIV>(tabv, t, G.T, G.Atop[t], ibase, ls.khi, ls.klo, E, G.c4, ls.w_in, ls.D_in, uabove, s - 2, Psi, cand);
        }
        for (int i = s + 1 + tid; i < N; i += kT) Psi[i] = 0.;                 // Numerov.h:427-428
        for (int o = 32; o > 0; o >>= 1) cand = max(cand, __shfl_xor(cand, o, 64));
        if (lane == 0) sh.red[par & 1][wv] = cand;
        __syncthreads();
        int mp = -1;
#pragma unroll
        for (int q = 0; q < kW; ++q) mp = max(mp, sh.red[par & 1][q]);
        ++par;
        // index 1 (after the last step): u_1 against u_2
        const double f1 = f_of_row(1), f2 = f_of_row(2);
        const double u1 = ls.w_fin / (1. - kInv12 * f1), u2 = (ls.w_fin - ls.D_fin) / (1. - kInv12 * f2);
        if (mp < 0 && (u1 < u2 || fabs(u1) > 1E15)) mp = 1;
        if (mp < 0) mp = 2;                                                    // Numerov.h:441: the loop never broke
        mo.matchPoint = mp;
        if (tid == 0) { Psi[s] = ls.us; Psi[s - 1] = ls.us1; }               // Numerov.h:435,441 (the two start values as they are)
    }
    const int mp = mo.matchPoint;
    if (mp < 2 || mp > s - 2) { mo.bad = 1; return mo; }
    __syncthreads();
    const double u_in_mp = Psi[mp];                                            // the inward value at the match point
    __syncthreads();
    // ---- outward: Psi[0] = 0, Psi[1] = GetBoundaryValueZero (Numerov.h:462-468); steps 1 .. mp-1 in ascending thread order
    double u_out_mp;
    {
        const int t = tid, ibase = t << G.logC;
        const int klo = max(1 - ibase, 0), khi = min(mp - 1 - ibase, C - 1);
        const bool has = khi >= klo;
        const double2 m = mm[t];
        const double Atop = G.Atop[t];
        const double xb = (Atop * fmax(fabs(m.x - E), fabs(m.y - E)) + G.c4) * kInv12;
        const int myvar = (t == 0) ? VDIV : (xb <= kX4 ? V4 : (xb <= kX8 ? V8 : VDIV));
        int var = V4;
        if (__any(has && myvar != V4)) var = __any(has && myvar == VDIV) ? VDIV : V8;
        Mat M = {1., 0., 0., 1.};
        double wd = 0, Dd = 0;
        if (has) {
            if (var == V4) match_out_rows<V4, false>(tabv, t, G.T, Atop, ibase, klo, khi, E, G.c4, M, wd, Dd, Psi);
            else if (var == V8) match_out_rows<V8, false>(tabv, t, G.T, Atop, ibase, klo, khi, E, G.c4, M, wd, Dd, Psi);
            else match_out_rows<VDIV, false>(tabv, t, G.T, Atop, ibase, klo, khi, E, G.c4, M, wd, Dd, Psi);
        }
        Mat P = M;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            Mat Ee;
            Ee.a = shfl_up_d(P.a, off); Ee.b = shfl_up_d(P.b, off); Ee.c = shfl_up_d(P.c, off); Ee.d = shfl_up_d(P.d, off);
            if (lane >= off) P = mat_mul(P, Ee);
        }
        const unsigned pp = par++ & 1;
        if (lane == 63) sh.wave_tot[pp][wv] = P;
        __syncthreads();
        const double f1 = f_of_row(1);
        const double w1 = (1. - kInv12 * f1) * zero1;                          // Numerov.h:467
        double w_in = w1, D_in = w1, w_fin, D_fin;
        {
            double w = w1, D = w1;
#pragma unroll
            for (int q = 0; q < kW; ++q) {
                if (q == wv) { w_in = w; D_in = D; }
                const Mat Wq = sh.wave_tot[pp][q];
                const double w2 = fma(Wq.a, w, Wq.b * D), D2 = fma(Wq.c, w, Wq.d * D);
                w = w2; D = D2;
            }
            w_fin = w; D_fin = D;
        }
        {
            Mat X;
            X.a = shfl_up_d(P.a, 1); X.b = shfl_up_d(P.b, 1); X.c = shfl_up_d(P.c, 1); X.d = shfl_up_d(P.d, 1);
            if (lane == 0) { X.a = 1; X.b = 0; X.c = 0; X.d = 1; }
            const double w2 = fma(X.a, w_in, X.b * D_in), D2 = fma(X.c, w_in, X.d * D_in);
            w_in = w2; D_in = D2;
        }
        if (has) {
            double w = w_in, D = D_in;
            if (var == V4) match_out_rows<V4, true>(tabv, t, G.T, Atop, ibase, klo, khi, E, G.c4, M, w, D, Psi);
            else if (var == V8) match_out_rows<V8, true>(tabv, t, G.T, Atop, ibase, klo, khi, E, G.c4, M, w, D, Psi);
            else match_out_rows<VDIV, true>(tabv, t, G.T, Atop, ibase, klo, khi, E, G.c4, M, w, D, Psi);
        }
        const double fmp = f_of_row(mp);
        u_out_mp = w_fin / (1. - kInv12 * fmp);                                // Numerov.h:489-492: the outward value at the match point
        mo.bad |= !(fabs(w_fin) < INFINITY) || !(fabs(D_fin) < INFINITY);
        if (tid == 0) { Psi[0] = 0.; Psi[1] = zero1; Psi[mp] = u_out_mp; }
    }
    const double factor = u_out_mp / u_in_mp;                                  // Numerov.h:494-498
    __syncthreads();
    // ---- Psi[i > mp] *= factor, then NormalizeNonUniform: Psi *= e^{i delta/2}, 1 / sqrt(Simpson38(1, Psi^2 Rp delta e^{delta i}))
    if (!fused_norm) {
        for (int i = mp + 1 + tid; i <= s; i += kT) Psi[i] *= factor;
        return mo;
    }
    double s1 = 0, s2 = 0, ends = 0;
    for (int i = tid; i < N; i += kT) {
        double p = Psi[i];
        if (i > mp) p *= factor;
        p *= eh[i];
        Psi[i] = p;
        const double r2 = p * p * cnst[i];
        if (i == 0 || i == N - 1) ends += r2;
        else if (i % 3 == 0) s2 += r2;
        else s1 += r2;
    }
    double part = ends + 3. * s1 + 2. * s2;                                    // Integral.h:50-73, summed in parallel
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    const unsigned pp = par++ & 1;
    if (lane == 0) sh.wave_tot[pp][wv].a = part;
    __syncthreads();
    double tot = 0;
#pragma unroll
    for (int q = 0; q < kW; ++q) tot += sh.wave_tot[pp][q].a;
    const double unorm = 1. / sqrt(tot * (3. / 8.));
    mo.bad |= !(unorm < INFINITY) || !(unorm > 0.);
    for (int i = tid; i < N; i += kT) Psi[i] *= unorm;
    return mo;
}

// per slot and lane: {min, max} of veff over the lane's rows
__global__ void __launch_bounds__(kT) k_scan_lane_minmax(const double* __restrict__ tabs, int N, int logC, double2* __restrict__ mm)
{
    const double* p = tabs + (size_t)blockIdx.x * N + threadIdx.x;      // row k of lane t at (k << kLogT) + t
    const int C = 1 << logC;
    double lo = INFINITY, hi = -INFINITY;
    bool nan = false;
    for (int k = 0; k < C; ++k) {
        const double v = p[(size_t)k << kLogT];
        nan = nan || (v != v);
        lo = fmin(lo, v); hi = fmax(hi, v);
    }
    double2 o;
    o.x = nan ? -INFINITY : lo; o.y = nan ? INFINITY : hi;
    mm[(size_t)blockIdx.x * kT + threadIdx.x] = o;
}

__global__ void __launch_bounds__(kT) k_scan_sweeps(ScanGrid G0, const double* __restrict__ gr, const double* __restrict__ gAtop, const double* __restrict__ gT,
                                                     int kind, const double* __restrict__ tabs, const double2* __restrict__ mms, const int* __restrict__ trial_slot,
                                                     const double* __restrict__ E, const int* __restrict__ limit, int* __restrict__ count,
                                                     double* __restrict__ u0, int* __restrict__ start, int* __restrict__ trip, int* __restrict__ bad)
{
    __shared__ ScanShared sh;
    ScanGrid G = G0;
    G.r = gr; G.Atop = gAtop; G.T = gT;           // kernel-argument pointers: global address space, scalar loads where the index is uniform
    const int q = blockIdx.x;
    const double* tab = tabs + (size_t)trial_slot[q] * G.N;
    const double2* mm = mms + (size_t)trial_slot[q] * kT;
    SweepOut o;
    if (kind == DFTA_SWEEP_COUNT) o = scan_sweep<DFTA_SWEEP_COUNT>(G, tab, mm, E[q], limit[q], sh, 0);
    else o = scan_sweep<DFTA_SWEEP_ZERO>(G, tab, mm, E[q], 0, sh, 0);
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


__global__ void __launch_bounds__(kT) k_scan_levels(ScanGrid G0, const double* __restrict__ gr, const double* __restrict__ gAtop, const double* __restrict__ gT,
                                                     dfta::Job* __restrict__ jobs, const int* __restrict__ chain_off, int chained,
                                                     const double* __restrict__ tabs, const double2* __restrict__ mms, int fixed_point,
                                                     unsigned long long* __restrict__ counters, ScanMatchArgs ma)
{
    __shared__ ScanShared sh;
    ScanGrid G = G0;
    G.r = gr; G.Atop = gAtop; G.T = gT;
    const int c = blockIdx.x;
    const size_t rows = (size_t)G.N;
    double handed = 0;
    bool have_handed = false;
    for (int k = chain_off[c]; k < chain_off[c + 1]; ++k) {
        dfta::Job* J = jobs + k;
        if (J->frozen) { if (ma.mode > 0 && threadIdx.x == 0) ma.jstart_keep[k] = -1; continue; }
        const int nodes = J->nodes, slot = J->slot;
        const double* tab = tabs + (size_t)slot * rows;
        const double2* mm = mms + (size_t)slot * kT;
        unsigned par = 0;
        int hint = 0;
        const double bottom0 = (chained && have_handed) ? handed : J->bottom0;
        int n_count = 0, n_zero = 0, bad = 0, len2 = 0, n_fixed = 0;
        long long pts = 0;
        // first bisection: the lowest energy with more than `nodes` nodes (DFTAtom.cpp:568-585)
        double toe = 50., boe = bottom0;
        // ma.mode < 0: the search is a PREDICTOR of the exact kernels' first spines (levels.hip): it brackets TopEnergy to a quarter of the
        // band the prediction is trusted to anyway and stops -- centre in J->top, half width in J->bottom (a copy of the records).  A
        // predictor need not sit on the reference's midpoints: with an SCF history it first probes the two ends of the history bracket
        // (plan_round's: hist_c +- (hist_w + 1e-10 |T| + 64e-12)) and bisects what is left -- ~20 sweeps instead of ~40 at steady state.
        if (ma.mode < 0 && J->hist_ok >= 2 && J->hist_d[0] >= 0) {
            const double T = J->hist_c[0], mh = J->hist_w[0] + 1e-10 * fabs(T) + 64 * kErr;
            // ... and once that bracket is within 2^12 x the scan's own band (twelve decisions) the prediction is not worth its sweeps: a scan
            // sweep occupies a compute unit for 50 us -- two exact trial lanes' worth of machine time -- and a decision moved from a tree to
            // the spine saves about two lanes.  The level keeps its history bracket (Z = 1..86 to its end: the predictor gains 10 - 25 ms per
            // step up to step 15, breaks even at 16 - 24 and lost 3 - 8 ms per step from there on before this rule).
            if (mh <= 4096. * (6e-11 * fabs(T) + 6e-10)) {
                if (threadIdx.x == 0) J->top = NAN;
                continue;
            }
            for (int side = 0; side < 2; ++side) {
                const double E = side ? T + mh : T - mh;
                if (!(E > boe && E < toe)) continue;
                const SweepOut o = scan_sweep<DFTA_SWEEP_COUNT>(G, tab, mm, E, nodes, sh, par++ & 1, hint);
                hint = o.start;
                bad |= o.bad;
                if (o.count > nodes) toe = E; else boe = E;
            }
        }
        while (toe - boe > (ma.mode < 0 ? fmax(kErr, 0.25 * (6e-11 * fmin(fabs(toe), fabs(boe)) + 6e-10)) : kErr)) {
            const double E = (toe + boe) / 2;
            const SweepOut o = scan_sweep<DFTA_SWEEP_COUNT>(G, tab, mm, E, nodes, sh, par++ & 1, hint);
            hint = o.start;
            ++n_count; bad |= o.bad;
            pts += o.start - 1 - (o.iexit > 0 ? o.iexit : 1);
            if (o.count > nodes) toe = E; else boe = E;
        }
        if (ma.mode < 0) {
            if (threadIdx.x == 0) { J->top = 0.5 * (toe + boe); J->bottom = 0.5 * (toe - boe); if (bad) atomicOr(counters + 3, 1ull); }
            continue;
        }
        const double top = toe;
        // second bisection: the lowest energy with at least `nodes` nodes (DFTAtom.cpp:587-603).  "count < 0" never holds: the
        // path of a node-less level is arithmetic (counted, not integrated -- as in levels.hip)
        boe = bottom0;
        while (toe - boe > kErr) {
            const double E = (toe + boe) / 2;
            ++n_count; ++len2;
            if (nodes == 0) { toe = E; continue; }
            const SweepOut o = scan_sweep<DFTA_SWEEP_COUNT>(G, tab, mm, E, nodes, sh, par++ & 1, hint);
            hint = o.start;
            bad |= o.bad;
            pts += o.start - 1 - (o.iexit > 0 ? o.iexit : 1);
            if (o.count < nodes) boe = E; else toe = E;
        }
        const double bottom = toe;
        // third bisection: the sign change of u(0) inside [bottom, top] (DFTAtom.cpp:513-534)
        double Top = top, Bot = bottom;
        int iter3 = 0, conv = 0, fixed = 0;
        double last_ad = 0;
        {
            const SweepOut o = scan_sweep<DFTA_SWEEP_ZERO>(G, tab, mm, Bot, 0, sh, par++ & 1, hint);
            hint = o.start;
            ++n_zero; bad |= o.bad; pts += o.start - 2;
            const bool sgnBottom = o.u0 > 0;
            while (iter3 < kIter3) {
                const double E = (Top + Bot) / 2;
                const SweepOut z = scan_sweep<DFTA_SWEEP_ZERO>(G, tab, mm, E, 0, sh, par++ & 1, hint);
                hint = z.start;
                ++n_zero; ++iter3; bad |= z.bad; pts += z.start - 2;
                const double Top_was = Top, Bot_was = Bot;
                if ((z.u0 > 0) == sgnBottom) Bot = E; else Top = E;
                const double ad = fabs(z.u0);
                last_ad = ad;
                if (Top - Bot < kErr && !(ad != ad) && ad < 1E15) { conv = 1; break; }
                if (fixed_point && Top == Top_was && Bot == Bot_was) {      // the same midpoint, sweep and decision to the cap (levels.hip)
                    const int rest = kIter3 - iter3;
                    n_zero += rest; n_fixed += rest; iter3 = kIter3; fixed = 1;
                    break;
                }
            }
        }
        handed = Bot - 3;                                            // DFTAtom.cpp:541
        have_handed = true;
        int matchPoint = 0;
        if (ma.mode) {                                               // "now really solve it" (DFTAtom.cpp:543-546)
            const int l = J->l;
            const double z1 = l == 0 ? ma.zero1[0] : (l == 1 ? ma.zero1[1] : (l == 2 ? ma.zero1[2] : ma.zero1[3]));
            const MatchOut mo = scan_match(G, tab, mm, Bot, z1, sh, par, hint, ma.Psi + (size_t)k * G.N, ma.eh, ma.cnst, ma.mode == 2);
            bad |= mo.bad;
            matchPoint = mo.matchPoint;
            pts += mo.start;                                         // inward from the cut-off to the match point + outward up to it
            if (threadIdx.x == 0) ma.jstart_keep[k] = mo.start;
        }
        if (threadIdx.x == 0) {
            if (ma.mode) J->matchPoint = matchPoint;
            J->top = top; J->bottom = bottom; J->toe = Top; J->boe = Bot; J->E = Bot;
            J->bottom0 = bottom0;
            J->status = conv ? DFTA_LEVEL_CONVERGED
                             : (DFTA_LEVEL_ITERATION_CAP | (fixed ? DFTA_LEVEL_FIXED_POINT : 0) | (!(last_ad < INFINITY) ? DFTA_LEVEL_U0_NONFINITE : 0));
            J->converged = conv; J->n_count = n_count; J->n_zero = n_zero; J->iter3 = iter3; J->n_fixed = n_fixed;
            J->cur_len[1] = len2; J->n_points = pts; J->phase = kPhDone; J->haveSgn = 1;
            const int skipped = (nodes == 0 ? len2 : 0) + n_fixed;          // counted, not integrated
            atomicAdd(counters, (unsigned long long)(n_count + n_zero - skipped));
            atomicAdd(counters + 1, (unsigned long long)pts);
            if (bad) atomicOr(counters + 3, 1ull);
        }
    }
}

// ---- the same search by a GROUP of K = 2^d - 1 workgroups per level: one round = the K midpoints of a depth-d bisection tree ----------
// One compute unit integrates a trial in 40 ... 80 us, and a single atom leaves 240 of 256 idle: the K members of a job each integrate one
// node of the tree that hangs at the running interval (member m: heap node m + 1, its midpoint found by following the node's path with
// the reference's (toe + boe) / 2), publish the outcome in ONE 64-bit word -- [round:16][trips:24][count:12][flags:12], an agent-scope
// atomic store; the word validates itself, no fence -- read the other members' words (agent-scope atomic loads, bounded spin) and all walk
// the tree with the reference's predicates: d decisions per round, ~150 / d rounds per level, the same decisions as one workgroup alone
// takes (tests: bit-identical job records).  A phase that ends inside a tree ends the round; the sweep at BottomEnergy that fixes the
// sign convention of the third bisection (DFTAtom.cpp:513) is a round of its own.  Member 0 then matches and normalises.
// The members wait for each other: the launch is cooperative (or plain under a profiler with the grid within the compute units), spins
// are bounded, a time-out sets counters[3] and the host repeats the solve on the exact kernels.
constexpr unsigned kNone = 1u << 4, kPos = 1u, kSmall = 2u, kNonFinite = 4u, kBad = 8u;

__device__ __forceinline__ unsigned long long xch_pack(unsigned round, int trips, int count, unsigned flags)
{
    return (static_cast<unsigned long long>(round & 0xffffu) << 48) | (static_cast<unsigned long long>(trips & 0xffffff) << 24) |
           (static_cast<unsigned long long>(count & 0xfff) << 12) | (flags & 0xfffu);
}

__global__ void __launch_bounds__(kT) k_scan_levels_group(ScanGrid G0, const double* __restrict__ gr, const double* __restrict__ gAtop, const double* __restrict__ gT,
                                                           dfta::Job* __restrict__ jobs, const int* __restrict__ live, int K, int spine,
                                                           const double* __restrict__ tabs, const double2* __restrict__ mms, int fixed_point,
                                                           unsigned long long* __restrict__ counters, unsigned long long* __restrict__ xch, ScanMatchArgs ma)
{
    __shared__ ScanShared sh;
    ScanGrid G = G0;
    G.r = gr; G.Atop = gAtop; G.T = gT;
    const int k = live[blockIdx.x / K], m = blockIdx.x % K;
    dfta::Job* J = jobs + k;
    const int nodes = J->nodes, slot = J->slot;
    const double* tab = tabs + (size_t)slot * G.N;
    const double2* mm = mms + (size_t)slot * kT;
    unsigned long long* X = xch + (size_t)(blockIdx.x / K) * 32;          // [parity][16]
    unsigned par = 0;
    int hint = 0;
    const double bottom0 = J->bottom0;
    // history bracket of the three bisections (set by the host from the previous two solves; hm < 0: none)
    const bool hok = J->hist_ok >= 2;
    // centre and half width of the history bracket as the host laid them out (levels.hip: T +- 2 |d|, or the extrapolated point)
    const double hT0 = J->hist_c[0], hT1 = J->hist_c[1], hT2 = J->hist_c[2];
    const double hm0 = hok && J->hist_d[0] >= 0 ? J->hist_w[0] + 1e-10 * fabs(hT0) + 64 * kErr : -1.0;
    const double hm1 = hok && J->hist_d[1] >= 0 ? J->hist_w[1] + 1e-10 * fabs(hT1) + 64 * kErr : -1.0;
    const double hm2 = hok && J->hist_d[2] >= 0 ? J->hist_w[2] + 1e-10 * fabs(hT2) + 64 * kErr : -1.0;
    int n_count = 0, n_zero = 0, bad = 0, len2 = 0, n_fixed = 0, iter3 = 0, conv = 0, fixed = 0, nonfinite = 0;
    long long pts = 0;
    // ph 1, 2: the two count bisections; 4: the sweep at BottomEnergy; 3: the u(0) bisection; 0: done
    int ph = 1;
    double lo = bottom0, hi = 50., top = 0, bottom = 0;
    bool sgnBottom = false;
    unsigned rnd = 0;
    while (ph) {
        ++rnd;                                                             // tags start at 1: 0 is the cleared state of the exchange words
        if (rnd >= 0xffffu) { if (threadIdx.x == 0) atomicOr(counters + 3, 3ull); return; }     // (a search takes ~40 rounds; the tag has 16 bits)
        // ---- the round's layout: a SPINE of s1 predicted decisions, one member each, and the full tree of depth d at its end.  The
        // prediction is the history bracket of the exact path (levels.h): this step's end point of the running bisection lies within
        // hm = 2 x the last movement of the previous step's; while a midpoint is outside [hT - hm, hT + hm] its decision is known
        // beforehand -- integrated at the reference's midpoint all the same, and checked: a miss ends the round after that decision.
        // Every member builds the same layout (no `break` in these loops: see the compiler note in DESIGN.md 4.2b).
        int s1 = 0, d = 0;
        unsigned spbits = 0;
        double sa = lo, sb = hi;
        int sit = iter3;
        if (ph != 4) {
            const double hT = ph == 1 ? hT0 : (ph == 2 ? hT1 : hT2), hm = ph == 1 ? hm0 : (ph == 2 ? hm1 : hm2);
            bool more = spine != 0 && hm >= 0;
            for (int j = 0; j < K; ++j) {
                if (more) {
                    if (ph == 3 ? sit >= kIter3 : !(sb - sa > kErr)) more = false;
                    else {
                        const double mid = (sb + sa) / 2;
                        const int pb = mid < hT - hm ? 1 : (mid > hT + hm ? 0 : -1);
                        if (pb < 0) more = false;
                        else {
                            spbits |= static_cast<unsigned>(pb) << s1;
                            ++s1; ++sit;
                            if (pb) sa = mid; else sb = mid;
                        }
                    }
                }
            }
            while ((2 << d) - 1 <= K - s1) ++d;                             // the largest tree that fits behind the spine: 2^d - 1 <= K - s1
        }
        // ---- this member's trial: spine node m, or heap node h = m - s1 + 1 of the tree at the spine's end
        unsigned long long mine = xch_pack(rnd, 0, 0, kNone);
        {
            double E = 0;
            bool have = false;
            if (ph == 4) { have = (m == 0); E = lo; }
            else if (m < s1) {
                double a = lo, b = hi;
                for (int j = 0; j < m; ++j) { const double mid = (b + a) / 2; if ((spbits >> j) & 1u) a = mid; else b = mid; }
                E = (b + a) / 2;
                have = true;
            } else {
                const int h = m - s1 + 1;
                if (h < (1 << d)) {
                    const int q = 31 - __clz(h);                           // depth of the node
                    double a = sa, b = sb;
                    int it = sit;
                    have = true;
                    for (int j = q - 1; j >= 0 && have; --j) {
                        if (ph == 3 ? it >= kIter3 : !(b - a > kErr)) { have = false; break; }
                        const double mid = (b + a) / 2;
                        if ((h >> j) & 1) a = mid; else b = mid;             // bit 1: "BottomEnergy = E"
                        ++it;
                    }
                    if (have && (ph == 3 ? it >= kIter3 : !(b - a > kErr))) have = false;
                    E = (b + a) / 2;
                }
            }
            if (have) {
                if (ph == 1 || ph == 2) {
                    const SweepOut o = scan_sweep<DFTA_SWEEP_COUNT>(G, tab, mm, E, nodes, sh, par++ & 1, hint);
                    hint = o.start;
                    mine = xch_pack(rnd, o.start - 1 - (o.iexit > 0 ? o.iexit : 1), o.count, o.bad ? kBad : 0u);
                } else {
                    const SweepOut z = scan_sweep<DFTA_SWEEP_ZERO>(G, tab, mm, E, 0, sh, par++ & 1, hint);
                    hint = z.start;
                    const double ad = fabs(z.u0);
                    mine = xch_pack(rnd, z.start - 2, 0, (z.u0 > 0 ? kPos : 0u) | ((!(ad != ad) && ad < 1E15) ? kSmall : 0u) | (!(ad < INFINITY) ? kNonFinite : 0u) | (z.bad ? kBad : 0u));
                }
            }
        }
        // ---- publish, gather
        if (threadIdx.x == 0) {
            unsigned long long* slotp = X + (rnd & 1) * 16;
            __hip_atomic_store(slotp + m, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // ONE wall-clock budget for the whole gather of a round (a bound per member added up to many seconds before the
            // fall-back): two seconds of the 100 MHz clock -- a sweep takes 40 .. 700 us -- then the abort bit, which every member polls
            bool timeout = false;
            const long long t_gather = wall_clock64();
            for (int q = 0; q < K; ++q) {
                unsigned long long v = q == m ? mine : 0ull;
                long spins = 0;
                while (q != m) {
                    v = __hip_atomic_load(slotp + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((v >> 48) == (rnd & 0xffffu)) break;
                    if ((++spins & 255) == 0 && (wall_clock64() - t_gather > 200000000ll ||
                                                 (__hip_atomic_load(counters + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2ull))) { timeout = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (timeout) break;
                sh.xres[q] = v;
            }
            if (timeout) { atomicOr(counters + 3, 3ull); sh.xres[0] = ~0ull; }
        }
        __syncthreads();
        if (sh.xres[0] == ~0ull) return;                                 // a member is missing: the host repeats the solve on the exact kernels
        // ---- walk the tree with the reference's predicates (every member alike)
        if (ph == 4) {
            const unsigned long long v = sh.xres[0];
            ++n_zero; pts += (v >> 24) & 0xffffff; bad |= (v & kBad) != 0;
            sgnBottom = (v & kPos) != 0;
            ph = 3;
        } else {
            bool stop = false;
            // one decision from the result word of the member that integrated the midpoint of (lo, hi); returns the bit taken
            auto take = [&](const unsigned long long v) -> int {
                const unsigned fl = static_cast<unsigned>(v & 0xfff);
                if (fl & kNone) { bad = 1; stop = true; return 0; }        // cannot happen: an active node without a result
                const int cnt = static_cast<int>((v >> 12) & 0xfff);
                const double E = (hi + lo) / 2;
                pts += (v >> 24) & 0xffffff; bad |= (fl & kBad) != 0;
                bool bit;
                if (ph == 1) { ++n_count; bit = !(cnt > nodes); }         // DFTAtom.cpp:579-582
                else if (ph == 2) { ++n_count; ++len2; bit = cnt < nodes; }   // DFTAtom.cpp:597-600
                else {
                    ++n_zero; ++iter3;
                    bit = ((fl & kPos) != 0) == sgnBottom;                // DFTAtom.cpp:522-525
                    nonfinite = (fl & kNonFinite) != 0;
                }
                const double hi_was = hi, lo_was = lo;
                if (bit) lo = E; else hi = E;
                if (ph == 3) {
                    const double width = hi - lo;
                    if (width < kErr && (fl & kSmall) != 0u) { conv = 1; stop = true; }
                    else if (fixed_point && hi == hi_was && lo == lo_was) {
                        const int rest = kIter3 - iter3;
                        n_zero += rest; n_fixed += rest; iter3 = kIter3; fixed = 1;
                        stop = true;
                    }
                }
                return bit ? 1 : 0;
            };
            for (int j = 0; j < s1; ++j) {                                 // the spine: a miss ends the round (the decision itself stands)
                if (!stop) {
                    if (ph == 3 ? iter3 >= kIter3 : !(hi - lo > kErr)) stop = true;
                    else {
                        const int bit = take(sh.xres[j]);
                        if (bit != static_cast<int>((spbits >> j) & 1u)) stop = true;
                    }
                }
            }
            int h = 1;
            while (h < (1 << d) && !stop) {
                if (ph == 3 ? iter3 >= kIter3 : !(hi - lo > kErr)) { stop = true; continue; }
                const int bit = take(sh.xres[s1 + h - 1]);
                h = 2 * h + bit;
            }
            // end of a phase?
            if (ph == 1 && !(hi - lo > kErr)) {
                top = hi; lo = bottom0; ph = 2;
                // ma.mode < 0: the search is a PREDICTOR for the exact kernels' first spines (levels.hip, LEVELS_SCAN_PREDICT): it ends with the
                // first bisection -- every member reaches this point in the same round -- and leaves TopEnergy in the record
                if (ma.mode < 0) { if (m == 0 && threadIdx.x == 0) { J->top = top; J->bottom = 0.0; if (bad) atomicOr(counters + 3, 1ull); } return; }
                if (nodes == 0) {                                          // "count < 0" never holds: arithmetic (levels.hip)
                    while (hi - lo > kErr) { hi = (hi + lo) / 2; ++n_count; ++len2; }
                }
            }
            if (ph == 2 && !(hi - lo > kErr)) { bottom = hi; lo = bottom; hi = top; ph = 4; }
            else if (ph == 3 && (conv || fixed || iter3 >= kIter3)) ph = 0;
        }
#ifdef DFTA_SCAN_DEBUG
        if (threadIdx.x == 0 && blockIdx.x == 0) printf("rnd %u ph %d lo %.6f hi %.6f iter3 %d conv %d fixed %d nz %d nc %d bad %d x0 %llx\n", rnd, ph, lo, hi, iter3, conv, fixed, n_zero, n_count, bad, sh.xres[0]);
#endif
        __syncthreads();                                                   // xres is rewritten in the next round
    }
    if (m != 0) return;
    const double Bot = lo, Top = hi;
    int matchPoint = 0;
    if (ma.mode) {
        const int l = J->l;
        const double z1 = l == 0 ? ma.zero1[0] : (l == 1 ? ma.zero1[1] : (l == 2 ? ma.zero1[2] : ma.zero1[3]));
        const MatchOut mo = scan_match(G, tab, mm, Bot, z1, sh, par, hint, ma.Psi + (size_t)k * G.N, ma.eh, ma.cnst, ma.mode == 2);
        bad |= mo.bad;
        matchPoint = mo.matchPoint;
        pts += mo.start;
        if (threadIdx.x == 0) ma.jstart_keep[k] = mo.start;
    }
    if (threadIdx.x == 0) {
        if (ma.mode) J->matchPoint = matchPoint;
        J->top = top; J->bottom = bottom; J->toe = Top; J->boe = Bot; J->E = Bot;
        J->status = conv ? DFTA_LEVEL_CONVERGED : (DFTA_LEVEL_ITERATION_CAP | (fixed ? DFTA_LEVEL_FIXED_POINT : 0) | (nonfinite ? DFTA_LEVEL_U0_NONFINITE : 0));
        J->converged = conv; J->n_count = n_count; J->n_zero = n_zero; J->iter3 = iter3; J->n_fixed = n_fixed;
        J->cur_len[1] = len2; J->n_points = pts; J->phase = kPhDone; J->haveSgn = 1;
        const int skipped = (nodes == 0 ? len2 : 0) + n_fixed;
        atomicAdd(counters, (unsigned long long)(n_count + n_zero - skipped));
        atomicAdd(counters + 1, (unsigned long long)pts);
        if (bad) atomicOr(counters + 3, 1ull);
    }
}

// interleaved tolerance-mode table of every slot: rows V + c_l
__global__ void k_scan_build_tab(double* __restrict__ tabs, const double* __restrict__ V, const double* __restrict__ cl,
                                 const int* __restrict__ slot_v, const int* __restrict__ slot_l, int N, int logC)
{
    const int slot = blockIdx.y;
    const int v = slot_v[slot], l = slot_l[slot];
    const int C = 1 << logC;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const size_t pos = (i == N - 1) ? (size_t)(N - 1) : (((size_t)(i & (C - 1)) << kLogT) + (i >> logC));
        tabs[(size_t)slot * N + pos] = V[(size_t)v * N + i] + cl[(size_t)l * N + i];
    }
}

ScanGrid scan_grid_of(const dfta_grid* g, const dfta_scan_tables& tb)
{
    ScanGrid G;
    G.N = g->N;
    G.logC = g->levels - kLogT;
    G.delta = g->delta;
    G.far_thr = g->far_arg_threshold;
    G.c4 = g->delta2p4;
    G.Rp = g->Rp;
    G.r = g->d_r;
    G.Atop = tb.Atop;
    G.T = tb.T;
    return G;
}

}  // namespace

int dfta_scan_supported(const dfta_grid* g) { return g && !g->uniform && g->levels >= 12 && g->levels <= 20; }

void dfta_scan_tables_destroy(dfta_scan_tables* tb)
{
    if (!tb) return;
    for (void* q : {(void*)tb->tabv_alloc, (void*)tb->mm, (void*)tb->Atop, (void*)tb->T_alloc}) if (q) (void)hipFree(q);
    *tb = dfta_scan_tables();
}

int dfta_scan_tables_create(dfta_ctx* ctx, const dfta_grid* g, int nslots, dfta_scan_tables* tb)
{
    *tb = dfta_scan_tables();
    tb->nslots = nslots;
    const int N = g->N, logC = g->levels - kLogT, C = 1 << logC;
    // A_i = 2 Rp^2 delta^2 exp(2 i delta) (Numerov.h:100) = Atop[t] T[k] for i = t C + k: the top row of every lane from the grid's own
    // exp table (host libm), the C ratios exp(-2 delta (C-1-k)) likewise
    std::vector<double> Atop(kT + 1), T(C);
    for (int t = 0; t < kT; ++t) Atop[t] = 2. * g->Rp2delta2 * g->h_e2[(size_t)t * C + C - 1];
    Atop[kT] = 2. * g->Rp2delta2 * g->h_e2[N - 1];
    for (int k = 0; k < C; ++k) T[k] = exp(-g->twodelta * static_cast<double>(C - 1 - k));
    constexpr size_t kPadTab = (size_t)kScanPadRows * kT, kPadT = 8;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&tb->tabv_alloc), sizeof(double) * ((size_t)nslots * N + kPadTab));
    if (e == hipSuccess) e = hipMemset(tb->tabv_alloc, 0, sizeof(double) * kPadTab);
    if (e == hipSuccess) tb->tabv = tb->tabv_alloc + kPadTab;
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&tb->mm), sizeof(double2) * (size_t)nslots * kT);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&tb->Atop), sizeof(double) * (kT + 1));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&tb->T_alloc), sizeof(double) * (C + kPadT));
    if (e == hipSuccess) e = hipMemset(tb->T_alloc, 0, sizeof(double) * kPadT);
    if (e == hipSuccess) tb->T = tb->T_alloc + kPadT;
    if (e == hipSuccess) e = hipMemcpy(tb->Atop, Atop.data(), sizeof(double) * (kT + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(tb->T, T.data(), sizeof(double) * C, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        snprintf(ctx->err, sizeof(ctx->err), "scan tables: %s", hipGetErrorString(e));
        dfta_scan_tables_destroy(tb);
        return DFTA_ERR_HIP;
    }
    return DFTA_OK;
}

int dfta_launch_scan_build_tab(dfta_ctx* ctx, const dfta_grid* g, const dfta_scan_tables& tb, const double* dV, const int* d_slot_v, const int* d_slot_l)
{
    hipLaunchKernelGGL(k_scan_build_tab, dim3(std::min(256, (g->N + 255) / 256), tb.nslots), dim3(256), 0, ctx->stream, tb.tabv, dV, g->d_cl,
                       d_slot_v, d_slot_l, g->N, g->levels - kLogT);
    DFTA_CHECK_LAUNCH(ctx);
    hipLaunchKernelGGL(k_scan_lane_minmax, dim3(tb.nslots), dim3(kT), 0, ctx->stream, tb.tabv, g->N, g->levels - kLogT, tb.mm);
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

int dfta_launch_scan_sweeps(dfta_ctx* ctx, const dfta_grid* g, int kind, int ntrials, const dfta_scan_tables& tb, const int* d_trial_slot,
                            const double* dE, const int* dLimit, int* dCount, double* dU0, int* dStart, int* dTrip, int* dBad)
{
    hipLaunchKernelGGL(k_scan_sweeps, dim3(ntrials), dim3(kT), 0, ctx->stream, scan_grid_of(g, tb), g->d_r, tb.Atop, tb.T, kind, tb.tabv, tb.mm, d_trial_slot, dE, dLimit, dCount, dU0,
                       dStart, dTrip, dBad);
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

int dfta_launch_scan_levels(dfta_ctx* ctx, const dfta_grid* g, dfta::Job* d_jobs, const int* d_chain_off, int nchains, int chained,
                            const dfta_scan_tables& tb, int fixed_point, unsigned long long* d_counters, int match_mode, double* d_Psi, int* d_jstart_keep)
{
    ScanMatchArgs ma;
    ma.mode = match_mode; ma.Psi = d_Psi; ma.eh = g->d_eh; ma.cnst = g->d_cnst; ma.jstart_keep = d_jstart_keep;
    for (int q = 0; q < 4; ++q) ma.zero1[q] = g->zero1[q];
    hipLaunchKernelGGL(k_scan_levels, dim3(nchains), dim3(kT), 0, ctx->stream, scan_grid_of(g, tb), g->d_r, tb.Atop, tb.T, d_jobs, d_chain_off, chained, tb.tabv, tb.mm, fixed_point, d_counters, ma);
    DFTA_CHECK_LAUNCH(ctx);
    return DFTA_OK;
}

#ifdef DFTA_SCAN_PROF
extern "C" int dfta_debug_scan_prof(unsigned long long* out16, int reset)
{
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_scan_prof), sizeof(unsigned long long) * 16) != hipSuccess) return DFTA_ERR_HIP;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_scan_prof), z, sizeof(z)) != hipSuccess) return DFTA_ERR_HIP; }
    return DFTA_OK;
}
#endif

int dfta_launch_scan_levels_group(dfta_ctx* ctx, const dfta_grid* g, dfta::Job* d_jobs, const int* d_live, int nlive, int K, const dfta_scan_tables& tb,
                                  int fixed_point, unsigned long long* d_counters, unsigned long long* d_xch, int match_mode, double* d_Psi, int* d_jstart_keep)
{
    ScanMatchArgs ma;
    ma.mode = match_mode; ma.Psi = d_Psi; ma.eh = g->d_eh; ma.cnst = g->d_cnst; ma.jstart_keep = d_jstart_keep;
    for (int q = 0; q < 4; ++q) ma.zero1[q] = g->zero1[q];
    int depth = dfta_knob("SCAN_NOSPINE") ? 0 : 1;         // (kernel argument `spine`) rounds start with a spine of predicted decisions
    DFTA_HIP(ctx, hipMemsetAsync(d_xch, 0, sizeof(unsigned long long) * 32 * (size_t)nlive, ctx->stream));
    ScanGrid G = scan_grid_of(g, tb);
    const double *pr = g->d_r, *pA = tb.Atop, *pT = tb.T, *ptab = tb.tabv;
    const double2* pmm = tb.mm;
    // the members of a job wait for each other: co-residency is the launch's business.  Under a profiler (rocprofiler-sdk 7.2 crashes in an
    // exit handler after a cooperative launch, see poisson.hip) the launch is an ordinary one: the grid is within the compute units, one
    // workgroup of 8 waves x 248 VGPRs fills a compute unit, and the bounded spins catch what is left.
    const bool plain = getenv("ROCP_TOOL_LIBRARIES") != nullptr || dfta_knob("SCAN_PLAIN_LAUNCH") != nullptr;
    if (plain) {
        hipLaunchKernelGGL(k_scan_levels_group, dim3(nlive * K), dim3(kT), 0, ctx->stream, G, pr, pA, pT, d_jobs, d_live, K, depth, ptab, pmm, fixed_point, d_counters, d_xch, ma);
        DFTA_CHECK_LAUNCH(ctx);
        return DFTA_OK;
    }
    void* args[] = {&G, &pr, &pA, &pT, &d_jobs, &d_live, &K, &depth, &ptab, &pmm, &fixed_point, &d_counters, &d_xch, &ma};
    const hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_scan_levels_group), dim3(nlive * K), dim3(kT), args, 0, ctx->stream);
    if (e != hipSuccess) { (void)hipGetLastError(); return DFTA_ERR_NOT_CONVERGED; }      // not co-resident: the caller falls back to one workgroup per level
    return DFTA_OK;
}

// ---- C ABI: the tolerance-mode twin of dfta_numerov_sweeps -----------------------------------------------------------------
extern "C" int dfta_numerov_sweeps_scan(dfta_ctx* ctx, const dfta_grid* g, int kind, int nV, const double* V, int ntrials, const int* vidx,
                                        const int* l, const double* E, const int* nodesLimit, int* count_out, double* u0_out,
                                        int* start_out, int* trip_out, int* fallback_out)
{
    if (!ctx || !g) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, dfta_scan_supported(g), "the scan sweeps need a logarithmic grid of 12 .. 20 multigrid levels");
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
    DFTA_HIP(ctx, dV.alloc((size_t)nV * N));
    struct Tables { dfta_scan_tables t; ~Tables() { dfta_scan_tables_destroy(&t); } } tb;
    int rc = dfta_scan_tables_create(ctx, g, nslots, &tb.t);
    if (rc) return rc;
    DFTA_HIP(ctx, hipMemcpyAsync(dV.p, V, (size_t)nV * N * sizeof(double), hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, dE.alloc(ntrials)); DFTA_HIP(ctx, dU0.alloc(ntrials));
    DFTA_HIP(ctx, dSv.alloc(nslots)); DFTA_HIP(ctx, dSl.alloc(nslots)); DFTA_HIP(ctx, dTs.alloc(ntrials)); DFTA_HIP(ctx, dLim.alloc(ntrials));
    DFTA_HIP(ctx, dCount.alloc(ntrials)); DFTA_HIP(ctx, dStart.alloc(ntrials)); DFTA_HIP(ctx, dTrip.alloc(ntrials)); DFTA_HIP(ctx, dBad.alloc(ntrials));
    DFTA_HIP(ctx, hipMemcpyAsync(dE.p, E, sizeof(double) * ntrials, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipMemcpyAsync(dSv.p, slot_v.data(), sizeof(int) * nslots, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipMemcpyAsync(dSl.p, slot_l.data(), sizeof(int) * nslots, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipMemcpyAsync(dTs.p, tslot.data(), sizeof(int) * ntrials, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipMemcpyAsync(dLim.p, lim.data(), sizeof(int) * ntrials, hipMemcpyHostToDevice, st));
    rc = dfta_launch_scan_build_tab(ctx, g, tb.t, dV.p, dSv.p, dSl.p);
    if (rc) return rc;
    DFTA_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    rc = dfta_launch_scan_sweeps(ctx, g, kind, ntrials, tb.t, dTs.p, dE.p, dLim.p, dCount.p, dU0.p, dStart.p, dTrip.p, dBad.p);
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
