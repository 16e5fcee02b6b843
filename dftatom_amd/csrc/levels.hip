// levels.hip -- device-side eigenvalue search for a batch of (potential, n, l) levels.
//
// Replaces DFTAtom::LoopOverLevels, LocateInterval, NormalizeNonUniform and the accumulation
// newDensity += occ * Psi^2 (DFTAtom.cpp:36-56, 493-604).
//
// The reference finds each eigenvalue with three sequential bisections (node count > n: top of the band;
// node count < n: bottom of the band; sign of u(0): the eigenvalue), ~140 sweeps per level, each sweep a
// sequential O(N) recurrence.  Here every bisection is a SPECULATIVE TREE: in one round the 2^d - 1
// possible midpoints of the next d bisection steps of every level are integrated concurrently (one lane
// per trial, numerov.hip), then one thread per chain walks its tree with the reference's predicates.
// The midpoints are generated with the reference's expression (toe + boe) / 2 along the same paths, so the
// walk reproduces the reference's decision sequence; ~53 sequential sweeps per phase become ceil(53/d) rounds.
// A tree may hang at the end of a SPINE, a predicted decision path that costs one trial per decision; predictions come
// from the previous SCF step, from the sibling level (the second bisection evaluates the predicate of the sibling's
// first one), from the position of the sign change of u(0) (upper end of the band for l = 0, scouted for l > 0).
// They select which midpoints are integrated speculatively and never enter a decision (k_plan, plan_round).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "internal.h"
#include "levels.h"
#include "ordered_sum.h"

namespace {

constexpr double kEnergyErr = 1E-12;   // DFTAtom.cpp:348
constexpr int kMaxIter3 = 500;         // DFTAtom.cpp:517

enum Phase { PH_WAIT = 0, PH_TOP = 1, PH_BOTTOM = 2, PH_ZERO = 3, PH_DONE = 4 };

// ---- trial layout of a round -------------------------------------------------------------------------------------
// Trial 0 of a job is the Bottom probe of the sign bisection (DFTAtom.cpp:513).  Trials 1..S are the SPINE: the
// bisection nodes along the predicted decision string (previous SCF step).  Trials S+g, g = 1..2^d'-1, form a full
// binary tree (heap order) rooted at the end of the spine.  Every node's energy is produced by the reference's own
// expression (toe + boe) / 2 applied along its path, so whichever nodes the walk visits carry exactly the energies
// the sequential bisection would have used; the prediction only selects WHICH nodes are integrated speculatively.
__device__ __forceinline__ int phase_index(int phase) { return phase - 1; }   // PH_TOP, PH_BOTTOM, PH_ZERO -> 0, 1, 2

__device__ __forceinline__ int tree_depth_for(int tpj, int spine)
{
    const int room = tpj - spine;            // heap indices 1 .. room-1 are available
    return room >= 2 ? 31 - __clz(room) : 0; // full levels: nodes 1 .. 2^d' - 1
}

__device__ __forceinline__ int capz_of(const dfta::Job& j, int tpj) { return j.capz > 0 ? j.capz : tpj; }

// the part of the band (count == nodes) that is certain so far, from the running first bisections of the job and its sibling
__device__ __forceinline__ void band_of(const dfta::Job& j, const dfta::Job* __restrict__ jobs, double& blo, double& bhi)
{
    blo = j.bottom0;
    if (j.nodes > 0) {
        blo = 1e300;                                                   // unknown: no band
        if (j.sib >= 0) {
            const dfta::Job sb = jobs[j.sib];
            if (sb.phase == PH_TOP) blo = sb.toe;                      // count > nodes-1 there
            else if (sb.phase == PH_BOTTOM || sb.phase == PH_ZERO || sb.phase == PH_DONE) blo = sb.top;
        }
    }
    bhi = j.phase == PH_TOP ? j.boe : j.top - kEnergyErr;             // count <= nodes there
}

__device__ __forceinline__ bool pred_bit(const dfta::Job& j, int ph, int k)
{
    if (j.use_sp) return (j.sp_bits >> (k & 63)) & 1ull;
    return (j.pred_bits[ph] >> (k & 63)) & 1ull;
}

// ---- expand: trial energies of the current round of every job, plus their far boundary values ----------------------
__global__ __launch_bounds__(256) void k_expand(const dfta::Job* __restrict__ jobs, const int* __restrict__ wave_job, int wshift, int ntrials, const double* __restrict__ r,
                                                int N, double delta, double far_thr, double* __restrict__ E,
                                                int* __restrict__ limit, int* __restrict__ start, double* __restrict__ us,
                                                double* __restrict__ us1, int* __restrict__ wave_kind,
                                                unsigned long long* __restrict__ issued, int uniform, double Rmax, double hstep)
{
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    if (gt >= ntrials) return;       // ntrials is a multiple of 64: whole waves leave
    const int job = wave_job[gt >> wshift];   // one entry per 64-trial block (wshift 6) or per trial (packed rounds: wshift 0)
    if (job < 0) {                    // slots that no job owns this round
        E[gt] = 0; limit[gt] = 0; start[gt] = 0;
        if ((gt & 63) == 0) wave_kind[gt >> 6] = DFTA_SWEEP_ZERO;
        return;
    }
    const dfta::Job j = jobs[job];
    const int tpj = j.tcap;
    const int h = gt - j.tbase;
    bool active = false;
    double e = 0;
    const int capz = capz_of(j, tpj);
    if (h >= capz) {
        // eigenvalue scouts: u(0) on a grid inside the band, later inside the bracket of its sign change
        if (j.phase == PH_TOP || j.phase == PH_BOTTOM) {
            const int n = tpj - capz, i = h - capz;
            if (j.se_state == 1) {
                e = j.se_lo + (i + 1) * ((j.se_hi - j.se_lo) / (n + 1));
                active = e > j.se_lo && e < j.se_hi;
            } else {
                // scan of the band: uniform in the middle, geometric towards both edges (1.5 bits per sample)
                double blo, bhi;
                band_of(j, jobs, blo, bhi);
                const int q = n / 4 < 32 ? n / 4 : 32;           // 48 bits towards each edge
                double u;
                if (i < q) u = exp2(-1.5 * (q - i) - 1.);
                else if (i >= n - q) u = 1. - exp2(-1.5 * (i - (n - q) + 1) - 1.);
                else u = 0.25 + (i - q + 1) * (0.5 / (n - 2 * q + 1));
                e = blo + u * (bhi - blo);
                active = bhi > blo && e > blo && e < bhi;
            }
        }
    } else if (j.phase == PH_TOP || j.phase == PH_BOTTOM || j.phase == PH_ZERO) {
        if (h == 0) {
            if (j.phase == PH_ZERO && !j.haveSgn) { active = true; e = j.boe; }   // DFTAtom.cpp:513
        } else {
            const int ph = phase_index(j.phase);
            const int S = j.spine;
            double hi = j.toe, lo = j.boe;
            int depth;                                        // decisions taken before this node in this round
            bool exists = true;
            if (h <= S) {
                depth = h - 1;
                for (int k = 0; k < depth; ++k) {
                    const double m = (hi + lo) / 2;
                    if (pred_bit(j, ph, j.phase_done + k)) lo = m; else hi = m;
                }
            } else {
                // the tree's trials are laid out in ENERGY order (in-order rank of the heap node): the 64 trials of a
                // sweep block then have neighbouring energies -- similar cut-off radii, and whole blocks above the count
                // threshold leave CountNodes early and free their compute unit
                const int rank = h - S;
                const int dsub = tree_depth_for(capz, S);
                int g = 1, gdepth = 0;
                exists = dsub > 0 && rank < (1 << dsub);
                if (exists) {
                    const int tz = __ffs(rank) - 1;
                    gdepth = dsub - 1 - tz;
                    g = (1 << gdepth) + (rank >> (tz + 1));
                }
                for (int k = 0; k < S; ++k) {
                    const double m = (hi + lo) / 2;
                    if (pred_bit(j, ph, j.phase_done + k)) lo = m; else hi = m;
                }
                for (int b = gdepth - 1; b >= 0; --b) {
                    const double m = (hi + lo) / 2;
                    if ((g >> b) & 1) lo = m; else hi = m;       // child 2g: toe = E ; child 2g+1: boe = E
                }
                depth = S + gdepth;
            }
            e = (hi + lo) / 2;
            if (j.phase == PH_ZERO) active = exists && (depth < kMaxIter3 - j.iter3);
            else active = exists && (hi - lo > kEnergyErr);       // loop condition of DFTAtom.cpp:571,589
        }
    }
    E[gt] = e;
    limit[gt] = j.nodes;
    int st = 0;
    if (active && uniform) {
        // uniform grid (Numerov.h:32-35,43-56,274-296): start at min(Rmax, 200 / sqrt(2|E|)), index (long)(startPoint / h)
        const double s = sqrt(2. * fabs(e));
        const double mr = 200. / s;
        const double sp = mr < Rmax ? mr : Rmax;
        st = static_cast<int>(static_cast<long>(sp / hstep));
        us[gt] = exp(-sp * s);
        us1[gt] = exp(-(sp - hstep) * s);
    } else if (active) {
        // GetMaxRadiusIndex (Numerov.h:119-136); exp(arg) < 1e-200 <=> arg < far_thr
        const double s = sqrt(2. * fabs(e));
        int maxIndex = N - 1, minIndex = 1;
        while (maxIndex - minIndex > 1) {
            const int mid = (maxIndex + minIndex) / 2;
            const double arg = -r[mid] * s - static_cast<double>(mid) * delta * 0.5;
            if (arg < far_thr) maxIndex = mid; else minIndex = mid;
        }
        st = maxIndex;
        us[gt] = exp(-r[st] * s - static_cast<double>(st) * delta * 0.5);              // Numerov.h:107
        us1[gt] = exp(-r[st - 1] * s - static_cast<double>(st - 1) * delta * 0.5);
    }
    start[gt] = st;
    if ((gt & 63) == 0) wave_kind[gt >> 6] = (j.phase == PH_ZERO || h >= capz) ? DFTA_SWEEP_ZERO : DFTA_SWEEP_COUNT;
    if (issued) {
        const unsigned long long m = __ballot(active);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(issued, (unsigned long long)__popcll(m));
    }
}

// ---- walk: follow each job's spine + tree with the reference's predicates ---------------------------------------------
// Cursor over the trial layout of a round: returns the trial index of the node reached after the decisions taken so
// far, or -1 when that node was not part of this round (prediction missed or tree exhausted).
struct Cursor {
    int S, dsub, k, g;     // spine length, subtree depth, decisions taken this round, heap index inside the subtree
    bool off;
    __device__ void init(int spine, int tpj) { S = spine; dsub = tree_depth_for(tpj, spine); k = 0; g = 1; off = false; }
    __device__ int node() const
    {
        if (off) return -1;
        if (k < S) return k + 1;
        const int gdepth = 31 - __clz(g);
        if (gdepth >= dsub) return -1;
        const int p = g - (1 << gdepth);
        return S + ((2 * p + 1) << (dsub - 1 - gdepth));      // in-order rank of heap node g (see k_expand)
    }
    __device__ void advance(bool bit, bool predicted)
    {
        if (k < S) { if (bit != predicted) off = true; }
        else g = 2 * g + (bit ? 1 : 0);
        ++k;
    }
};

__device__ __forceinline__ void record_bit(dfta::Job& j, int ph, bool bit)
{
    if (j.phase_done < 64) {
        if (bit) j.cur_bits[ph] |= (1ull << j.phase_done);
        j.cur_len[ph] = j.phase_done + 1;
    }
    ++j.phase_done;
}

// at the end of a phase: how long did the prediction hold?  Plan the first spine of the next phase.
__device__ __forceinline__ void finish_phase(dfta::Job& j, int ph)
{
    int common = 0;
    const int m = j.cur_len[ph] < j.pred_len[ph] ? j.cur_len[ph] : j.pred_len[ph];
    while (common < m && (((j.cur_bits[ph] ^ j.pred_bits[ph]) >> common) & 1ull) == 0ull) ++common;
    j.trust[ph] = common;       // read by the NEXT solve (after pred := cur)
    j.phase_done = 0;
    j.miss = 0;
}

// Simulate the running bisection from its current interval against a bracket [blo, bhi] of the energy at which its
// predicate flips: a midpoint at or below blo takes the "boe = E" branch (bit 1), one at or above bhi the other one.
__device__ __forceinline__ void predict_from_bracket(dfta::Job& j, double lo, double hi, double blo, double bhi, bool strict_end)
{
    unsigned long long bits = 0;
    int k = j.phase_done;
    while (k < 64 && (strict_end ? !(hi - lo < kEnergyErr) : (hi - lo > kEnergyErr))) {
        const double m = (hi + lo) / 2;
        bool bit;
        if (m <= blo) bit = true;
        else if (m >= bhi) bit = false;
        else break;
        if (bit) { bits |= 1ull << k; lo = m; } else hi = m;
        ++k;
    }
    j.sp_bits = bits;
    j.sp_len = k;
}

// The band of energies around a transition inside which CountNodes' count and the sign of u(0) are round-off (not monotonic in
// E).  Spines stop there -- a predicted decision inside it is a coin toss, and a miss forfeits the round's tree.  Measured on
// Rn's fifteen levels (16384-point scans of the count and of sign u(0) around every transition, end of round 2): the band is
// 1.2e-12 .. 7.4e-12 |E| wide (6p: 1.4e-11 |E| = 2.5e-12 absolute), the same for both predicates.  Guards: rel |E| + abs on either
// side of a predicted transition (sibling's end point, own top for l = 0), g_secant_noise |E| added to the secant's error bound.
// Round 1 used 2e-11 / 64e-12 / 3e-11; DFTA_LEVELS_NOISE="rel,abs,secant" overrides (read when a solver is created).
__device__ double g_noise_rel = 1e-11, g_noise_abs = 16e-12, g_secant_noise = 1.5e-11;
__device__ int g_fixed_point = 1;      // 0 ($DFTA_DEBUG LEVELS_NOFIXEDPOINT): a third bisection on its fixed point is integrated to the iteration cap (tests)
// Spine of the job's next round.  `jobs` is read for the sibling only (k_plan runs after every walk of the round).
__device__ __forceinline__ void plan_round(dfta::Job& j, const dfta::Job* __restrict__ jobs, int tpj)
{
    const int ph = phase_index(j.phase);
    // A miss ON the spine forfeits the tree of that round, so a spine from the previous SCF step stops one bit short of
    // what held last time (eigenvalues move by roughly half as much every SCF step: the prediction gains about one bit
    // per step anyway).
    int S = j.trust[ph] - 1 - j.phase_done;
    const int avail = j.pred_len[ph] - j.phase_done;
    if (S > avail) S = avail;
    if (j.hist_ok == 2) S = 0;                 // two steps of history: the bracket below replaces the bit-prefix rule
    j.use_sp = 0;
    j.sp_len = 0;
    // history bracket (levels.h): this step's end point within kHistSafety x the last movement of the previous one
    constexpr double kHistSafety = 2.0;
    unsigned long long hbits = 0;
    int hlen = 0;
    if (j.hist_ok == 2 && !j.miss && j.hist_d[ph] >= 0) {
        const double T = j.hist_T[ph];
        const double m = kHistSafety * j.hist_d[ph] + 1e-10 * fabs(T) + 64 * kEnergyErr;
        predict_from_bracket(j, j.boe, j.toe, T - m, T + m, j.phase == PH_ZERO);
        hbits = j.sp_bits;
        hlen = j.sp_len;
        j.sp_len = 0;
    }
    // the last decisions before the predicted flip are left to the tree: at that scale (a few 1e-11) the counted nodes
    // and the sign of u(0) are not monotonic in the energy, and a miss on the spine costs the whole round
    const double kGuard = g_noise_abs;
    const double kNoise = g_noise_rel;
    bool secant = false;
    if (j.phase == PH_TOP) {
        if (j.sc_ok && j.miss < 2) {
            predict_from_bracket(j, j.boe, j.toe, j.sc_lo - kGuard, j.sc_hi + kGuard, false);
            secant = true;
        }
    } else if (j.phase == PH_BOTTOM) {
        if (j.nodes == 0) predict_from_bracket(j, j.boe, j.toe, -1e300, -1e300, false);   // "count < 0" never holds
        else if (j.sib >= 0) {
            const dfta::Job sb = jobs[j.sib];
            if (sb.phase == PH_BOTTOM || sb.phase == PH_ZERO || sb.phase == PH_DONE) {
                // the spine stops where the count stops being a monotonic function of the energy (round-off of the sweep,
                // about 1e-11 |E|): a spine that runs into that band misses and forfeits the round's tree
                const double gd = kGuard + kNoise * fabs(sb.top);
                predict_from_bracket(j, j.boe, j.toe, sb.top - gd, sb.top + gd, false);
            }
            else if (sb.phase == PH_TOP)
                predict_from_bracket(j, j.boe, j.toe, sb.boe - kGuard, sb.toe + kGuard, false);
        }
    } else if (j.phase == PH_ZERO) {
        // l == 0: no inner turning point, the count changes exactly where u(0) changes sign -- at the upper end
        // (inside ~2e-11 |E| of it the sign of u(0), like the count, is round-off: a spine that runs into that band misses and
        // forfeits the round's tree)
        if (j.l == 0) { const double gd = kGuard + kNoise * fabs(j.top); predict_from_bracket(j, j.boe, j.toe, j.top - gd, j.top + gd, true); }
        else if (j.se_state == 1 && j.se_tok && !j.miss) predict_from_bracket(j, j.boe, j.toe, j.se_tlo - kGuard, j.se_thi + kGuard, true);
        else if (j.se_state == 1) predict_from_bracket(j, j.boe, j.toe, j.se_lo - kGuard, j.se_hi + kGuard, true);
    }
    // scouts for the third bisection of l > 0 (whole blocks of their own kind: tpj >= 128)
    j.capz = tpj;
    if (j.l > 0 && tpj >= 128 && (j.phase == PH_TOP || j.phase == PH_BOTTOM) && !j.se_stop &&
        !(j.phase == PH_TOP && j.phase_done == 0) &&
        !(j.se_state == 1 && j.se_hi - j.se_lo <= kEnergyErr)) {
        double blo, bhi;
        band_of(j, jobs, blo, bhi);
        if (j.se_state == 1 || bhi > blo) j.capz = tpj / 2;
    }
    // a long predicted spine needs the whole tree behind it to finish the bisection in this round: the scouts pause
    if (j.sp_len - j.phase_done >= 40) j.capz = tpj;
    // the history bracket where nothing better (sibling, top rule, scouts, secant) reaches further
    if (hlen > j.sp_len) { j.sp_bits = hbits; j.sp_len = hlen; secant = false; }
    // once a prediction has missed in this phase the predicted path and the real one have parted: plain trees from there,
    // except for spines from the secant estimate, which is made afresh from this round's samples (until one of those misses)
    if (j.miss && !secant) { S = 0; j.sp_len = 0; }
    if (j.miss && secant) S = 0;
    if (j.sp_len - j.phase_done > S) { S = j.sp_len - j.phase_done; j.use_sp = secant ? 2 : 1; }
    if (S > j.capz / 2 - 1) S = j.capz / 2 - 1;  // keep at least half of the trials for the tree
    j.spine = S > 0 ? S : 0;
}

// Secant estimate of the end point of the first bisection from the samples kept in the job (see levels.h).  With
// F(E) = (stop index of the sample - stop index of the boe-side sample) + phi -- the distance of the nearest zero of u from
// the point where CountNodes stops, in grid cells -- the count changes where F crosses 0.  The error bound is the
// interpolation error of the secant with the second divided difference taken from the third sample (times 4).
__device__ double g_secant_kappa = 0.25;     // 0: secant only (DFTA_LEVELS_SECANT_KAPPA overrides, read when a solver is created)
__device__ __forceinline__ void secant_predict(dfta::Job& j, const double2* __restrict__ veff /* table rows of the job's slot */)
{
    j.sc_ok = 0;
    const int ia = j.sc_is[0], ib = j.sc_is[1], ic = j.sc_is[2];
    if (ia < 0 || ib < 0 || ic < 0 || (ia & kStopOver)) return;
    // x0(E) = stop index + phi = position of the zero of u next to the stop point, in grid cells (absolute): smooth in E
    // whatever made the sweep stop.  The count changes where x0(E) reaches the point at which CountNodes stops for THAT
    // energy: s(E) = 0 for l == 0, the inner turning point -- the largest i below the well with veff_i > E -- otherwise,
    // a step function that comes down as E goes up (by about a cell while the zero moves a few dozen).
    auto X = [&](int k) { return static_cast<double>(j.sc_is[k] & ~kStopOver) + j.sc_phi[k]; };
    const double a = j.sc_e[0], b = j.sc_e[1], c = j.sc_e[2];
    const double xa = X(0), xb = X(1), xc = X(2);
    if (!(b > a) || c == a || c == b || !(xb > xa) || !(fabs(xc) < 1e300) || !(xa < ia)) return;
    const double w = b - a;
    const double f1 = (xb - xa) / w;
    const double f2 = ((xc - xb) / (c - b) - f1) / (c - a);
    double t = 0;
    bool found = false, at_step = false;
    int s = ia;                                       // stop index at E = a (the boe-side sample ran to its turning point)
    for (int it = 0; it < 16 && !found; ++it) {
        // the stop index stays s while E < veff_s (l == 0: for ever)
        const double Ej = (ia != 0 && s >= 1) ? veff[s].x : 1e300;
        const double tl = a + (static_cast<double>(s) - xa) / f1;        // the zero reaches s here
        if (tl < Ej) { t = tl; found = true; }
        else if (xa + (Ej - a) * f1 >= static_cast<double>(s - 1)) { t = Ej; found = true; at_step = true; }   // the stop point
        else --s;                                                                 // steps over the zero at E = veff_s
        if (s < 1 && ia != 0) break;
    }
    if (!found || !(t > a && t < b)) return;
    // 4 x |f2 / f1| (b - a)^2 / 4 for the secant, plus the scale below which the count is no longer a monotonic function of
    // the energy (round-off of the sweep: about 1e-11 of |E|)
    double e = (at_step ? 0.0 : fabs(f2 / f1) * w * w) + g_secant_noise * fabs(t);
    // The third sample gives more than a bound: the parabola through the three samples (Newton form
    // xa + f1 (E - a) + f2 (E - a)(E - b)) meets the stop point at t - f2 (t - a)(t - b) / f1 to first order -- measured, the
    // end point of the bisection sits at -0.20 .. -0.25 of the secant's error bound, i.e. ON that correction, time after
    // time.  The corrected estimate is trusted to a fraction g_secant_kappa of the correction itself (plus the noise floor);
    // it is dropped where it would cross the turning-point step that the secant's solution lies under.
    if (!at_step && g_secant_kappa > 0) {
        double tq = t;
        for (int it = 0; it < 2; ++it) tq = a + (static_cast<double>(s) - xa - f2 * (tq - a) * (tq - b)) / f1;
        const double Ej = (ia != 0 && s >= 1) ? veff[s].x : 1e300;
        if (tq > a && tq < b && tq < Ej) {
            const double e2 = g_secant_kappa * fabs(tq - t) + g_secant_noise * fabs(tq);
            if (e2 < e) { e = e2; t = tq; }
        }
    }
    if (!(e < w * 0.125)) return;
    j.sc_lo = t - e;
    j.sc_hi = t + e;
    j.sc_ok = 1;
}

__device__ void walk_job(dfta::Job& j, const int* __restrict__ count, const double* __restrict__ u0, const double* __restrict__ phi,
                         const int* __restrict__ istop, const int* __restrict__ trip, const double2* __restrict__ veff)
{
    const int tpj = j.tcap, base = j.tbase;
    Cursor c;
    c.init(j.spine, capz_of(j, tpj));
    if (j.phase == PH_TOP) {                                        // DFTAtom.cpp:568-585
        double hi = j.toe, lo = j.boe;
        while (hi - lo > kEnergyErr) {
            const int h = c.node();
            if (h < 0) { if (c.off) j.miss = (j.use_sp == 2) ? 2 : (j.miss > 1 ? j.miss : 1); j.toe = hi; j.boe = lo; secant_predict(j, veff); return; }
            const double e = (hi + lo) / 2;
            const int cn = count[base + h];
            ++j.n_count;
            if (trip) j.n_points += trip[base + h];
            const bool bit = !(cn > j.nodes);                        // 1: boe = E
            if (bit) lo = e; else hi = e;
            {
                const int side = bit ? 0 : 1;
                j.sc_e[2] = j.sc_e[side]; j.sc_phi[2] = j.sc_phi[side]; j.sc_is[2] = j.sc_is[side];
                j.sc_e[side] = e; j.sc_phi[side] = phi[base + h]; j.sc_is[side] = istop[base + h];
            }
            c.advance(bit, pred_bit(j, 0, j.phase_done));
            record_bit(j, 0, bit);
        }
        finish_phase(j, 0);
        j.sc_ok = 0;
        j.top = hi;
        j.toe = hi;
        j.boe = j.bottom0;                                          // DFTAtom.cpp:587
        j.phase = PH_BOTTOM;
        if (j.nodes == 0) {
            // A level without nodes: the second bisection asks "count < 0", which CountNodes can never answer with yes
            // (Numerov.h:272-349 counts up from 0).  Its whole path -- toe = E at every step (DFTAtom.cpp:589-601) -- is
            // therefore known without a single sweep: it is taken here, in the round that ended the first bisection, and
            // only counted in n_count (the reference does integrate those ~52 trials).
            double hi2 = j.toe, lo2 = j.boe;
            while (hi2 - lo2 > kEnergyErr) {
                hi2 = (hi2 + lo2) / 2;
                ++j.n_count;
                record_bit(j, 1, false);
            }
            finish_phase(j, 1);
            j.bottom = hi2;                                         // BottomEnergy = toe
            j.toe = j.top;
            j.boe = hi2;
            j.haveSgn = 0;
            j.iter3 = 0;
            j.phase = PH_ZERO;
        }
        return;
    }
    if (j.phase == PH_BOTTOM) {                                     // DFTAtom.cpp:587-603
        double hi = j.toe, lo = j.boe;
        while (hi - lo > kEnergyErr) {
            const int h = c.node();
            if (h < 0) { if (c.off) j.miss = 1; j.toe = hi; j.boe = lo; return; }
            const double e = (hi + lo) / 2;
            const int cn = count[base + h];
            ++j.n_count;
            if (trip) j.n_points += trip[base + h];
            const bool bit = (cn < j.nodes);                         // 1: boe = E
            if (bit) lo = e; else hi = e;
            c.advance(bit, pred_bit(j, 1, j.phase_done));
            record_bit(j, 1, bit);
        }
        finish_phase(j, 1);
        j.bottom = hi;                                              // BottomEnergy = toe
        j.toe = j.top;
        j.boe = hi;
        j.haveSgn = 0;
        j.iter3 = 0;
        j.phase = PH_ZERO;
        return;
    }
    if (j.phase == PH_ZERO) {                                       // DFTAtom.cpp:513-534
        if (!j.haveSgn) {
            const double d0 = u0[base];
            ++j.n_zero;
            if (trip) j.n_points += trip[base];
            j.sgnBottom = d0 > 0;
            j.haveSgn = 1;
        }
        double hi = j.toe, lo = j.boe;
        bool conv = false, fixed = false;
        double last_ad = 0;
        while (j.iter3 < kMaxIter3) {
            const int h = c.node();
            if (h < 0) { if (c.off) j.miss = 1; j.toe = hi; j.boe = lo; return; }
            const double e = (hi + lo) / 2;
            const double d = u0[base + h];
            ++j.n_zero;
            if (trip) j.n_points += trip[base + h];
            ++j.iter3;
            const bool bit = ((d > 0) == (j.sgnBottom != 0));        // 1: BottomEnergy = E
            const double hi_was = hi, lo_was = lo;
            if (bit) lo = e; else hi = e;
            c.advance(bit, pred_bit(j, 2, j.phase_done));
            record_bit(j, 2, bit);
            const double ad = fabs(d);
            last_ad = ad;
            if (hi - lo < kEnergyErr && !isnan(ad) && ad < 1E15) { conv = true; break; }
            if (g_fixed_point && hi == hi_was && lo == lo_was) {
                // A level whose u(0) never gets below 1e15 (or is NaN) keeps the reference bisecting until its 500-iteration cap
                // (DFTAtom.cpp:517-534) although the interval has long collapsed: once a step leaves (toe, boe) as they were, the
                // midpoint, its sweep, its sign and the decision repeat unchanged to the end -- a fixed point.  The remaining
                // iterations are taken here without their sweeps (counted: the reference integrates every one of them); they used to
                // cost a round per ~7 of them -- 70 rounds, 2.1 s per SCF step at 1 048 577 nodes whenever a level ended that way.
                const int rest = kMaxIter3 - j.iter3;
                j.n_zero += rest;
                j.n_fixed += rest;
                j.iter3 = kMaxIter3;
                fixed = true;
                break;
            }
        }
        finish_phase(j, 2);
        j.toe = hi;
        j.boe = lo;
        j.E = lo;                                                    // level.E = BottomEnergy
        j.converged = conv ? 1 : 0;
        // how it ended (include/dftatom_hip.h): the reference folds all of it into didNotConverge (DFTAtom.cpp:517-539)
        j.status = conv ? DFTA_LEVEL_CONVERGED
                        : (DFTA_LEVEL_ITERATION_CAP | (fixed ? DFTA_LEVEL_FIXED_POINT : 0) | (!(last_ad < INFINITY) ? DFTA_LEVEL_U0_NONFINITE : 0));
        j.phase = PH_DONE;
        j.spine = 0;
        j.capz = 0;
    }
}

// ---- scouts: bracket of the sign change of u(0) inside the band, one wave per job, before the walk of the round ------
__global__ __launch_bounds__(64) void k_scout(dfta::Job* __restrict__ jobs, const double* __restrict__ E,
                                              const int* __restrict__ start, const double* __restrict__ u0)
{
    const int job = blockIdx.x, lane = threadIdx.x;
    const dfta::Job j = jobs[job];
    const int tpj = j.tcap;
    const int capz = capz_of(j, tpj);
    if (!(j.phase == PH_TOP || j.phase == PH_BOTTOM) || capz >= tpj) return;
    const int base = j.tbase;
    // samples [capz, tpj) in ascending energy, m consecutive ones per lane
    const int n = tpj - capz, m = n / 64;
    const bool bracketed = j.se_state == 1;
    bool all_active = true;
    int first = 0x7fffffff;                  // first sample whose sign differs from its left neighbour / from se_sl
    int prev;                                // sign of the sample to the left of this lane's first one (-1: none)
    {
        const int last = base + capz + lane * m + m - 1;
        const int mine = start[last] >= 2 ? (u0[last] > 0 ? 1 : 0) : -1;
        prev = __shfl_up(mine, 1);
        if (lane == 0) prev = bracketed ? j.se_sl : -1;
    }
    for (int q = 0; q < m; ++q) {
        const int i = lane * m + q, idx = base + capz + i;
        if (start[idx] < 2) { all_active = false; prev = -1; continue; }
        const int sg = u0[idx] > 0 ? 1 : 0;
        const bool flip = bracketed ? (sg != j.se_sl) : (prev >= 0 && sg != prev);
        if (flip && i < first) first = i;
        prev = sg;
    }
    for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off));
    all_active = __ballot(all_active) == ~0ull;
    if (lane != 0 || !all_active) return;
    const double* Es = E + base + capz;
    // u(0) itself is the smooth function here: every sweep starts from the analytic decaying solution, so its scale does
    // not depend on where the sweep starts
    const double* Ps = u0 + base + capz;
    double nlo, nhi, plo, phi_hi;
    if (!bracketed) {
        if (first >= n) return;              // first >= 1 here
        nlo = Es[first - 1]; plo = Ps[first - 1];
        nhi = Es[first]; phi_hi = Ps[first];
        jobs[job].se_sl = u0[base + capz + first - 1] > 0 ? 1 : 0;
        jobs[job].se_state = 1;
    } else {
        nlo = first < n ? (first > 0 ? Es[first - 1] : j.se_lo) : Es[n - 1];
        plo = first < n ? (first > 0 ? Ps[first - 1] : j.se_plo) : Ps[n - 1];
        nhi = first < n ? Es[first] : j.se_hi;
        phi_hi = first < n ? Ps[first] : j.se_phi;
        if (!((nhi - nlo) * 2 < j.se_hi - j.se_lo)) jobs[job].se_stop = 1;
    }
    jobs[job].se_lo = nlo;
    jobs[job].se_hi = nhi;
    jobs[job].se_plo = plo;
    jobs[job].se_phi = phi_hi;
    // secant estimate of the sign change from phi at the two ends, error bound from a third sample (the next one outside
    // the bracket on either side); speculation only: it predicts the third bisection (plan_round)
    int tok = 0;
    double tlo = 0, thi = 0;
    {
        double c = 0, pc = NAN;
        if (first < n && first + 1 < n) { c = Es[first + 1]; pc = Ps[first + 1]; }
        else if (first < n && first >= 2) { c = Es[first - 2]; pc = Ps[first - 2]; }
        const double w = nhi - nlo;
        if (w > 0 && plo * phi_hi < 0 && fabs(pc) < 1e300 && c != nlo && c != nhi) {
            const double f1 = (phi_hi - plo) / w;
            const double f2 = ((pc - phi_hi) / (c - nhi) - f1) / (c - nlo);
            const double t = nlo + w * (plo / (plo - phi_hi));
            const double e = fabs(f2 / f1) * w * w + g_secant_noise * fabs(t);
            if (e < w * 0.125) { tok = 1; tlo = t - e; thi = t + e; }
        }
    }
    jobs[job].se_tok = tok;
    jobs[job].se_tlo = tlo;
    jobs[job].se_thi = thi;
}

__global__ void k_walk(dfta::Job* __restrict__ jobs, const int* __restrict__ chain_off, int nchains,
                       const int* __restrict__ count, const double* __restrict__ u0, const double* __restrict__ phi,
                       const int* __restrict__ istop, const int* __restrict__ trip, const double2* __restrict__ tab, int N,
                       int* __restrict__ ndone)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchains) return;
    int done = 0;
    for (int k = chain_off[c]; k < chain_off[c + 1]; ++k) {
        dfta::Job j = jobs[k];
        if (j.phase == PH_DONE) { ++done; continue; }
        if (j.phase == PH_WAIT) {
            // DFTAtom.cpp:541: BottomEnergy = level.E - 3 handed to the next level of the chain
            const double bot = (k == chain_off[c]) ? j.bottom0 : jobs[k - 1].E - 3;
            j.bottom0 = bot;
            j.toe = 50;                                             // DFTAtom.cpp:499
            j.boe = bot;
            j.phase = PH_TOP;
            j.phase_done = 0;
            j.miss = 0;
            j.capz = 0;
            j.se_state = 0;
            j.se_stop = 0;
            j.se_tok = 0;
            j.sc_is[0] = j.sc_is[1] = j.sc_is[2] = -1;
            j.sc_ok = 0;
            jobs[k] = j;
            break;                                                  // its trials are generated next round
        }
        walk_job(j, count, u0, phi, istop, trip, tab + (size_t)j.slot * N);
        jobs[k] = j;
        if (j.phase == PH_DONE) { ++done; continue; }
        break;
    }
    if (done) atomicAdd(ndone, done);
}

// spines of the next round, after every walk of this one (a job reads its sibling's first-bisection result)
__global__ void k_plan(dfta::Job* __restrict__ jobs, int njobs, int nopredict, int tpj_override)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= njobs) return;
    dfta::Job j = jobs[k];
    if (j.phase != PH_TOP && j.phase != PH_BOTTOM && j.phase != PH_ZERO) return;
    const int tpj = tpj_override > 0 ? tpj_override : j.tcap;   // packed rounds: "with room to spare", k_pack cuts it to size
    plan_round(j, jobs, tpj);
    if (nopredict) { j.spine = 0; j.capz = tpj; j.use_sp = 0; j.sp_len = 0; j.sp_bits = 0; }     // plain trees only (DFTA_LEVELS_NOPREDICT)
    jobs[k].spine = j.spine;
    jobs[k].capz = j.capz;
    jobs[k].use_sp = j.use_sp;
    jobs[k].sp_bits = j.sp_bits;
    jobs[k].sp_len = j.sp_len;
}

// ---- packed rounds: many jobs (batches of atoms) -- the trials of a round laid out job after job, no 64-slot block per job ----
// A 64-trial block of the sweep kernels has ONE table slot (potential, l) and ONE kind (CountNodes / SolutionInZero); its
// lanes are independent otherwise.  With a block per job (the static layout) a job cannot have fewer than 64 trials, and the
// work of a step is the number of blocks: a depth-5 tree behind a 10-decision spine integrates 42 trials for 15 decisions
// and pays for 64.  Here the jobs of a round are grouped by (slot, kind) -- 1s..6s of one atom share a table, 2p..6p the
// next -- and laid out back to back inside their group, a job taking exactly probe + spine + 2^d - 1 trials; only the end
// of a group is padded to a block.  The depth d is the same for every job of a round and is chosen from what is still
// searching: the largest one whose trials fit the launch that fills the machine (latency regime: ~1.5 passes of one block
// per compute unit, the pipelined kernel; else two waves per SIMD of the fused one), never below dmin.  As jobs finish, the
// rest get deeper trees.  Which midpoints are integrated changes; the decisions do not (the walk follows the reference's
// predicates on whatever nodes it finds).
constexpr int kPackThreads = 1024;
constexpr int kPackSpineCap = 40;      // decisions of one round's spine (the tree behind it adds d more)

__device__ __forceinline__ bool job_searching(int phase) { return phase == PH_TOP || phase == PH_BOTTOM || phase == PH_ZERO; }

__global__ __launch_bounds__(kPackThreads) void k_pack(dfta::Job* __restrict__ jobs, int njobs, int nslots, const int* __restrict__ slot_off,
                                                       const int* __restrict__ slot_jobs, int lanes_small, int dsmall, int lanes_large, int dmin, int dmax,
                                                       int* __restrict__ gsz, int* __restrict__ goff, int* __restrict__ lane_job,
                                                       int* __restrict__ wave_slot, int* __restrict__ out)
{
    __shared__ long long s_sum[kPackThreads / 64];
    __shared__ int s_cnt[kPackThreads / 64];
    __shared__ int s_scan[kPackThreads];
    __shared__ int s_d;
    const int tid = threadIdx.x;
    // (1) what is still searching, and how long its spines are
    long long ss = 0;
    int a = 0;
    for (int k = tid; k < njobs; k += kPackThreads) {
        if (!job_searching(jobs[k].phase)) continue;
        int S = jobs[k].spine;
        if (S > kPackSpineCap) { S = kPackSpineCap; jobs[k].spine = S; }
        ss += S;
        ++a;
    }
    for (int off = 32; off > 0; off >>= 1) { ss += __shfl_xor(ss, off); a += __shfl_xor(a, off); }
    if ((tid & 63) == 0) { s_sum[tid >> 6] = ss; s_cnt[tid >> 6] = a; }
    __syncthreads();
    if (tid == 0) {
        long long S = 0;
        long long A = 0;
        for (int w = 0; w < kPackThreads / 64; ++w) { S += s_sum[w]; A += s_cnt[w]; }
        auto fit = [&](long long T) { int d = dmin; while (d < dmax && S + (A << (d + 1)) <= T) ++d; return d; };
        int d = fit(lanes_small);
        if (d < dsmall) d = fit(lanes_large);
        s_d = d;
        out[1] = d;
        out[2] = (int)A;
    }
    __syncthreads();
    const int d = s_d;
    // (2) size of every (slot, kind) group, padded to whole blocks
    for (int s = tid; s < nslots; s += kPackThreads) {
        int nc = 0, nz = 0;
        for (int q = slot_off[s]; q < slot_off[s + 1]; ++q) {
            const int k = slot_jobs[q];
            const int ph = jobs[k].phase;
            if (!job_searching(ph)) continue;
            const int t = jobs[k].spine + (1 << d);
            if (ph == PH_ZERO) nz += t; else nc += t;
        }
        gsz[2 * s] = (nc + 63) & ~63;
        gsz[2 * s + 1] = (nz + 63) & ~63;
    }
    __syncthreads();
    // (3) exclusive scan of the group sizes: consecutive chunks per thread, scan of the chunk sums in LDS
    const int n2 = 2 * nslots;
    const int chunk = (n2 + kPackThreads - 1) / kPackThreads;
    const int c0 = min(tid * chunk, n2), c1 = min(c0 + chunk, n2);
    int local = 0;
    for (int i = c0; i < c1; ++i) local += gsz[i];
    s_scan[tid] = local;
    __syncthreads();
    for (int off = 1; off < kPackThreads; off <<= 1) {
        const int v = tid >= off ? s_scan[tid - off] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    int run = s_scan[tid] - local;
    for (int i = c0; i < c1; ++i) { goff[i] = run; run += gsz[i]; }
    if (tid == kPackThreads - 1) out[0] = s_scan[tid];       // trials of the round (a multiple of 64)
    __syncthreads();
    // (4) the jobs of a group back to back.  What the padding of the group's last block leaves free goes to deeper trees, a
    // level at a time, to the job with the shallowest tree (among equals: the one furthest behind -- earlier bisection, wider
    // interval); only what is left after that belongs to nobody.
    for (int s = tid; s < nslots; s += kPackThreads) {
        const int q0 = slot_off[s], q1 = slot_off[s + 1];
        int used[2] = {0, 0};
        for (int q = q0; q < q1; ++q) {
            const int k = slot_jobs[q];
            const int ph = jobs[k].phase;
            if (!job_searching(ph)) { jobs[k].tbase = 0; jobs[k].tcap = 0; continue; }
            jobs[k].tcap = d;                                      // depth of the job's tree, for now
            used[ph == PH_ZERO] += jobs[k].spine + (1 << d);
        }
        for (int kind = 0; kind < 2; ++kind) {
            int room = gsz[2 * s + kind] - used[kind];
            while (room >= (1 << d)) {
                int best = -1, bd = 1 << 30, bph = 0;
                double bw = 0;
                for (int q = q0; q < q1; ++q) {
                    const int k = slot_jobs[q];
                    const int ph = jobs[k].phase;
                    if (!job_searching(ph) || (ph == PH_ZERO) != (kind == 1)) continue;
                    const int dj = jobs[k].tcap;
                    const double w = jobs[k].toe - jobs[k].boe;
                    if (dj < bd || (dj == bd && (ph < bph || (ph == bph && w > bw)))) { best = k; bd = dj; bph = ph; bw = w; }
                }
                if (best < 0 || (1 << bd) > room || bd >= dmax) break;
                room -= 1 << bd;
                jobs[best].tcap = bd + 1;
            }
        }
        int c = goff[2 * s], z = goff[2 * s + 1];
        const int cend = c + gsz[2 * s], zend = z + gsz[2 * s + 1];
        for (int b = c >> 6; b < (cend >> 6); ++b) wave_slot[b] = s;
        for (int b = z >> 6; b < (zend >> 6); ++b) wave_slot[b] = s;
        for (int q = q0; q < q1; ++q) {
            const int k = slot_jobs[q];
            const int ph = jobs[k].phase;
            if (!job_searching(ph)) continue;
            const int t = jobs[k].spine + (1 << jobs[k].tcap);
            jobs[k].tcap = t;
            jobs[k].capz = t;
            if (ph == PH_ZERO) { jobs[k].tbase = z; z += t; } else { jobs[k].tbase = c; c += t; }
        }
        for (; c < cend; ++c) lane_job[c] = -1;
        for (; z < zend; ++z) lane_job[z] = -1;
    }
}

// owner of every trial slot of a packed round (one wave per job)
__global__ __launch_bounds__(256) void k_pack_lanes(const dfta::Job* __restrict__ jobs, int njobs, int* __restrict__ lane_job)
{
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= njobs) return;
    if (!job_searching(jobs[k].phase)) return;
    const int base = jobs[k].tbase, cap = jobs[k].tcap;
    for (int i = lane; i < cap; i += 64) lane_job[base + i] = k;
}

// ---- latency mode: the trial slots of a round are re-allotted among the jobs that are still searching -----------------
// With a handful of jobs (one atom) a round is one pass of at most `budget / 64` blocks over the compute units whatever
// the blocks do, and the number of rounds is set by the slowest job: the levels whose last decisions cannot be predicted
// (the count / the sign of u(0) are not monotonic in E below ~1e-11 |E|: 7 decisions for an outer level, 14-16 for 2s / 1s)
// need one tree of that depth or several rounds of shallower ones.  So each round every active job asks for what it can
// use -- probe + spine + a tree as deep as the decisions that remain after the spine (6 .. kMaxTreeDepth), twice that when
// it also scouts -- and the requests are cut back, deepest tree first, until they fit.  Which midpoints are integrated
// changes, the decisions taken do not (the walk follows the reference's predicates on whatever nodes it finds).
constexpr int kMaxTreeDepth = 14;
__device__ __forceinline__ int decisions_left(const dfta::Job& j)
{
    double w = j.toe - j.boe;
    int n = 0;
    if (j.phase == PH_ZERO) { while (!(w < kEnergyErr) && n < 64) { w *= 0.5; ++n; } }
    else                    { while (w > kEnergyErr && n < 64) { w *= 0.5; ++n; } }
    return n;
}

__global__ __launch_bounds__(64) void k_allot(dfta::Job* __restrict__ jobs, int njobs, int budget, int nopredict, int* __restrict__ wave_job,
                                              int* __restrict__ wave_slot, const int* __restrict__ live)
{
    // `live` (may be null): the njobs <= 64 jobs of a larger list that are still to be solved (the rest are frozen: finished atoms
    // of an SCF batch); lane k then stands for job live[k]
    __shared__ int s_S[64], s_base[64], s_cap[64], s_id[64];
    const int k = threadIdx.x;            // njobs <= 64 in this mode
    const int kk = (live && k < njobs) ? live[k] : k;
    s_id[k] = kk;
    bool act = false;
    int S = 0, r = 0, sc = 0, rem = 0, left = 0;
    if (k < njobs) {
        dfta::Job j = jobs[kk];
        act = (j.phase == PH_TOP || j.phase == PH_BOTTOM || j.phase == PH_ZERO);
        if (act) {
            plan_round(j, jobs, 1 << 14);                      // what would it do with room to spare?
            if (nopredict) { j.spine = 0; j.capz = 1 << 14; }
            S = j.spine;
            sc = j.capz < (1 << 14);
            // a tree that ends the phase if ten levels can (1024 trials: what a job's share of a pass affords), else equal shares
            // of the rounds it takes anyway; deeper phase-ending trees are handed out below, from what the others leave
            left = decisions_left(j) - S;
            if (left <= 10) r = left < 6 ? 6 : left;
            else { const int nr = (left + 9) / 10; r = (left + nr - 1) / nr; if (r < 6) r = 6; }
            // rounds this job still has in front of it (a level's three bisections take about 3 + 2 + 2 rounds): the slots go to
            // the jobs that are furthest behind, because the round count of a step is theirs
            rem = (left + 9) / 10 + (j.phase == PH_TOP ? 4 : (j.phase == PH_BOTTOM ? 2 : 0));
        }
    }
    // One lane per job from here on (the kernel is one wave): selections are wave-wide maxima of a key, sums are wave sums.
    auto wave_max = [](int v) { for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off)); return v; };
    auto wave_sum = [](int v) { for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off); return v; };
    auto slots_at = [&](int depth) { const int t = 1 + S + (1 << depth); const int u = (t + 127) & ~127; return sc ? 2 * u : u; };
    int tot = wave_sum(act ? slots_at(r) : 0);
    // What the modest requests leave goes, furthest-behind job first, to deeper trees that SAVE A ROUND of the phase: the
    // smallest depth with ceil(left / depth) one less than now, up to kMaxTreeDepth levels.  (Round 2 first let every job ask
    // for its phase-ending tree outright: a level with 14 undecided decisions then asked for the whole pass, was cut to 13 --
    // two rounds all the same -- and the cuts took the ninth level from a job that needed exactly nine: an extra round for a
    // single decision.  And a level with 23 decisions left got three rounds of 8 where 12 + 11 were affordable.)
    {
        bool stuck = !act;
        for (int guard = 0; guard < 256; ++guard) {
            // furthest behind first (largest rem), then the fewest decisions left, then the lowest job
            const bool cand = act && !stuck && r < left;
            const int key = cand ? (((rem & 0x3fff) << 16) | ((255 - min(left, 255)) << 8) | (63 - k)) : -1;
            const int best = wave_max(key);
            if (best < 0) break;
            const int lane_pick = 63 - (best & 0xff);
            const int nr = (left + max(r, 1) - 1) / max(r, 1);                   // rounds of this phase at the present depth
            const int target = nr > 1 ? (left + nr - 2) / (nr - 1) : left;
            const int cost = (target <= kMaxTreeDepth && target > r) ? slots_at(target) - slots_at(r) : (1 << 29);
            const int c = __shfl(cost, lane_pick);
            const bool ok = tot + c <= budget;                                   // wave-uniform
            if (k == lane_pick) { if (ok) { r = target; --rem; } else stuck = true; }
            if (ok) tot += c;
        }
    }
    // still too much: cut the job that is least behind (then the deepest tree among equals), down to 8 levels first, then to 6
    for (int floor = 8; floor >= 6 && tot > budget; floor -= 2) {
        for (int guard = 0; guard < 1024 && tot > budget; ++guard) {
            const bool cand = act && r > floor;
            const int key = cand ? ((((0x3fff - (rem & 0x3fff))) << 16) | (r << 8) | (63 - k)) : -1;
            const int best = wave_max(key);
            if (best < 0) break;
            const int lane_pick = 63 - (best & 0xff);
            const int delta = slots_at(r) - slots_at(max(r - 1, 0));
            tot -= __shfl(delta, lane_pick);
            if (k == lane_pick) --r;
        }
    }
    // slots in job order; what does not fit any more (cuts exhausted) is clamped exactly as a sequential pass would
    {
        int cap = act ? slots_at(r) : 0;
        int base = cap;                                                          // inclusive prefix sum
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(base, off); if (k >= off) base += v; }
        base -= cap;
        if (tot > budget) {                                                      // rare: sequential semantics by one lane
            s_S[k] = cap;
            __syncthreads();
            if (k == 0) {
                int b = 0;
                for (int q = 0; q < njobs; ++q) {
                    int cq = s_S[q];
                    if (b + cq > budget) cq = ((budget - b) / 128) * 128;
                    s_base[q] = b;
                    s_cap[q] = cq;
                    b += cq;
                }
            }
            __syncthreads();
        } else {
            s_base[k] = base;
            s_cap[k] = cap;
            __syncthreads();
        }
    }
    // the block table of the round, by all lanes (a lone thread pays a memory round trip per entry)
    const int my_slot = k < njobs ? jobs[kk].slot : 0;
    s_S[k] = my_slot;                                          // (the spine lengths are not needed any more)
    __syncthreads();
    for (int w = k; w < (budget >> 6); w += 64) {
        int q = -1;
        for (int t = 0; t < njobs; ++t)
            if (w * 64 >= s_base[t] && w * 64 < s_base[t] + s_cap[t]) q = t;
        wave_job[w] = q >= 0 ? s_id[q] : -1;
        wave_slot[w] = q >= 0 ? s_S[q] : 0;
    }
    if (k < njobs) { jobs[kk].tbase = s_base[k]; jobs[kk].tcap = s_cap[k]; }
    if (k < njobs && act) {                                    // the plan for the slots the job really got
        dfta::Job j = jobs[kk];
        j.tbase = s_base[k];
        j.tcap = s_cap[k];
        if (j.tcap < 128) {                                    // nothing left for it this round (cannot happen while budget >= 128 njobs)
            jobs[kk].spine = 0; jobs[kk].capz = j.tcap; jobs[kk].use_sp = 0; jobs[kk].sp_bits = 0; jobs[kk].sp_len = 0;
            return;
        }
        plan_round(j, jobs, j.tcap);
        if (nopredict) { j.spine = 0; j.capz = j.tcap; j.use_sp = 0; j.sp_len = 0; j.sp_bits = 0; }
        jobs[kk].spine = j.spine;
        jobs[kk].capz = j.capz;
        jobs[kk].use_sp = j.use_sp;
        jobs[kk].sp_bits = j.sp_bits;
        jobs[kk].sp_len = j.sp_len;
    }
}

// ---- normalise (DFTAtom.cpp:36-56) ------------------------------------------------------------------------------
// one block per job; the first two waves perform the Simpson 3/8 sum in the reference's order (the other rules: the first wave)
constexpr int kNormThreads = 1024;     // the two pointwise passes are chains of memory round trips of a single block
__global__ __launch_bounds__(kNormThreads) void k_normalize(double* __restrict__ Psi, double* __restrict__ G, int N,
                                                            const double* __restrict__ eh, const double* __restrict__ cnst,
                                                            const int* __restrict__ jstart, double step, int rule)
{
    __shared__ __attribute__((aligned(16))) double lds[dfta::kTile];
    __shared__ double rtab[64];
    __shared__ double s_unorm;
    if (jstart && jstart[blockIdx.x] < 0) return;     // frozen job: its normalised Psi stands
    double* P = Psi + (size_t)blockIdx.x * N;
    double* g = G + (size_t)blockIdx.x * N;
    for (int i = threadIdx.x; i < N; i += kNormThreads) {
        const double p = P[i] * eh[i];            // Psi[i] *= exp(i * deltaGrid * 0.5)
        P[i] = p;
        double r2 = p * p;
        r2 *= cnst[i];                            // result2[i] *= Rp * deltaGrid * exp(deltaGrid * i)
        g[i] = r2;
    }
    __syncthreads();
    // Simpson38(1, .) on the logarithmic grid, Simpson38(h, .) on the uniform one (DFTAtom.cpp:27,51)
    if (rule == DFTA_INT_SIMPSON38) {
        const double integral = dfta::block_simpson38(g, N, step, lds, rtab);
        if (threadIdx.x == 0) s_unorm = 1. / sqrt(integral);
    } else if (threadIdx.x < 64) {
        const double integral = dfta::wave_integrate(rule, g, N, step, lds, rtab);
        if (threadIdx.x == 0) s_unorm = 1. / sqrt(integral);
    }
    __syncthreads();
    const double unorm = s_unorm;
    for (int i = threadIdx.x; i < N; i += kNormThreads) P[i] *= unorm;
}

// newDensity[v][i] += occ * Psi[i] * Psi[i] for i < N-1, levels of a potential in their order (DFTAtom.cpp:558-559)
__global__ void k_accumulate_density(const double* __restrict__ Psi, const dfta::Job* __restrict__ jobs,
                                     const int* __restrict__ v_off, int N, double* __restrict__ newDensity)
{
    const int v = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        double acc = newDensity[(size_t)v * N + i];
        if (i < N - 1)
            for (int k = v_off[v]; k < v_off[v + 1]; ++k) {
                const double p = Psi[(size_t)k * N + i];
                acc += jobs[k].occ * p * p;
            }
        newDensity[(size_t)v * N + i] = acc;
    }
}

// min_i Veff_l(i), i = 1..N-1, of every table slot (one block per slot; NaN entries are ignored)
__global__ __launch_bounds__(1024) void k_slot_min(const double2* __restrict__ tab, int N, double* __restrict__ slot_min)
{
    __shared__ double red[16];
    const double2* T = tab + (size_t)blockIdx.x * N;
    double m = INFINITY;
    for (int i = 1 + threadIdx.x; i < N; i += 1024) {
        const double v = T[i].x;
        if (v < m) m = v;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_xor(m, off);
        if (o < m) m = o;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = red[0];
        for (int w = 1; w < 16; ++w) r = fmin(r, red[w]);
        slot_min[blockIdx.x] = r;
    }
}

// BATCHED: every level starts un-chained from max(-Z^2-1, min Veff_l): no eigenvalue lies below the minimum of
// the effective potential, and below it the node count of l >= 1 misfires
__global__ void k_clamp_bottoms(dfta::Job* __restrict__ jobs, int njobs, const double* __restrict__ slot_min)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= njobs) return;
    const double vmin = slot_min[jobs[k].slot];
    const double b = (vmin > jobs[k].bottom0) ? vmin : jobs[k].bottom0;
    jobs[k].bottom0 = b;
    jobs[k].boe = b;
}

__global__ void k_job_energies(const dfta::Job* __restrict__ jobs, int njobs, double* __restrict__ E, int* __restrict__ slot,
                               int* __restrict__ l)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= njobs) return;
    E[k] = jobs[k].E;
    slot[k] = jobs[k].slot;
    l[k] = jobs[k].l;
}

__global__ void k_store_match(dfta::Job* __restrict__ jobs, int njobs, const int* __restrict__ mp, const int* __restrict__ jstart)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < njobs && !jobs[k].frozen) {
        jobs[k].matchPoint = mp[k];
        jobs[k].n_points += jstart[k];        // inward from the cut-off to the match point + outward up to it
    }
}

// Early match solves run on a SECOND stream while the next round runs on the first, whose k_walk rewrites the job records.  What the
// second stream needs of them is therefore SNAPSHOT on the first stream, in order with the round -- right after k_walk, before the
// event the second stream waits for: the eigenvalue of every job and whether its search has ended.  (Before round 4 the second stream
// read jobs[k].E and, two kernels later, jobs[k].phase: a level that finished in the NEXT round between those two reads would have
// been matched with a stale energy.  Unlikely -- a sweep outlasts three small kernels -- but not ordered.)
__global__ void k_snapshot_done(const dfta::Job* __restrict__ jobs, int njobs, double* __restrict__ snapE, int* __restrict__ snapReady)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= njobs) return;
    snapE[k] = jobs[k].E;
    snapReady[k] = jobs[k].phase == PH_DONE && !jobs[k].frozen;
}
// second stream: the snapshot's finished jobs that have not been matched yet take their energy; the others keep what they had
__global__ void k_take_ready(const double* __restrict__ snapE, const int* __restrict__ snapReady, const int* __restrict__ matched, int njobs,
                             double* __restrict__ jE, int* __restrict__ take)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= njobs) return;
    const int t = snapReady[k] && !matched[k];
    take[k] = t;
    if (t) jE[k] = snapE[k];
    else if (!(jE[k] == jE[k])) jE[k] = -1.;         // never written yet: any finite energy (its cut-off index is masked out below)
}
// of the jobs whose cut-off indices were just computed, keep the taken ones (the others get -1: skipped by k_match); remember the kept
// ones' cut-off index for the statistics and the final passes
__global__ void k_mask_ready(const int* __restrict__ take, int njobs, int* __restrict__ jstart, int* __restrict__ matched,
                             int* __restrict__ jstart_keep)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= njobs) return;
    if (take[k]) { matched[k] = 1; jstart_keep[k] = jstart[k]; }
    else jstart[k] = -1;
}
// slot and l of every job (they never change during a run)
__global__ void k_job_slots(const dfta::Job* __restrict__ jobs, int njobs, int* __restrict__ slot, int* __restrict__ l)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= njobs) return;
    slot[k] = jobs[k].slot;
    l[k] = jobs[k].l;
}
// the final pass: every job that is neither frozen nor matched already
__global__ void k_mask_rest(const dfta::Job* __restrict__ jobs, int njobs, int* __restrict__ jstart, const int* __restrict__ matched,
                            int* __restrict__ jstart_keep)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= njobs) return;
    if (jobs[k].frozen) { jstart[k] = -1; jstart_keep[k] = -1; }
    else if (matched[k]) jstart[k] = -1;
    else jstart_keep[k] = jstart[k];
}

// frozen jobs are skipped by the match / normalise kernels: their cut-off index is replaced by -1
__global__ void k_mask_frozen(const dfta::Job* __restrict__ jobs, int njobs, int* __restrict__ jstart)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < njobs && jobs[k].frozen) jstart[k] = -1;
}

}  // namespace

namespace dfta {

LevelSolver::~LevelSolver() { release(); }

void LevelSolver::release()
{
    void* ptrs[] = {d_jobs, d_chain_off, d_chain_off_b, d_v_off, d_slot_v, d_slot_l, d_tab, d_E, d_limit, d_start, d_us, d_us1, d_count,
                    d_u0, d_phi, d_istop, d_trip, d_wave_job, d_wave_kind, d_wave_slot, d_wave_first, d_wave_cnt, d_counters, d_Psi, d_Q, d_jE, d_jslot, d_jl,
                    d_jstart, d_jus, d_jus1, d_jmp, d_slot_min, d_bounds};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    dfta_scan_tables_destroy(&scan_tb);
    if (d_scan_live) (void)hipFree(d_scan_live);
    if (d_scan_xch) (void)hipFree(d_scan_xch);
    d_scan_live = nullptr; d_scan_xch = nullptr;
    if (d_jmatched) (void)hipFree(d_jmatched);
    if (d_jstart_keep) (void)hipFree(d_jstart_keep);
    d_jmatched = nullptr; d_jstart_keep = nullptr;
    for (int** q : {&d_lane_job, &d_slot_off, &d_slot_jobs, &d_gsz, &d_goff, &d_pack_out, &d_live}) { if (*q) (void)hipFree(*q); *q = nullptr; }
    if (st2) { (void)hipStreamDestroy(st2); st2 = nullptr; }
    if (ev_walk) { (void)hipEventDestroy(ev_walk); ev_walk = nullptr; }
    if (ev_early) { (void)hipEventDestroy(ev_early); ev_early = nullptr; }
    if (ev_taken) { (void)hipEventDestroy(ev_taken); ev_taken = nullptr; }
    for (void* q : {(void*)d_snapE, (void*)d_snapReady, (void*)d_jtake}) if (q) (void)hipFree(q);
    d_snapE = nullptr; d_snapReady = nullptr; d_jtake = nullptr;
    for (hipEvent_t& e : ev) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    d_jobs = nullptr; d_chain_off = nullptr; d_chain_off_b = nullptr; d_v_off = nullptr; d_slot_v = nullptr; d_slot_l = nullptr; d_tab = nullptr;
    d_E = nullptr; d_limit = nullptr; d_start = nullptr; d_us = nullptr; d_us1 = nullptr; d_count = nullptr; d_u0 = nullptr; d_phi = nullptr; d_istop = nullptr; d_trip = nullptr;
    d_wave_kind = nullptr; d_wave_slot = nullptr; d_wave_first = nullptr; d_wave_cnt = nullptr; d_counters = nullptr; d_wave_job = nullptr;
    d_Psi = nullptr; d_Q = nullptr; d_jE = nullptr; d_jslot = nullptr; d_jl = nullptr; d_jstart = nullptr; d_jus = nullptr;
    d_jus1 = nullptr; d_jmp = nullptr; d_slot_min = nullptr; d_bounds = nullptr;
}

// jobs must be ordered by potential index v (levels of one potential contiguous, in the reference's (N,L) order)
int LevelSolver::setup(dfta_ctx* c, const dfta_grid* grid, int mode_, int tree_depth, int nV_, const std::vector<JobSpec>& specs)
{
    release();
    ctx = c; g = grid; mode = mode_; nV = nV_;
    use_prediction = dfta_knob("LEVELS_NOPREDICT") == nullptr;   // measurements / tests: every spine and scout off
    {   // tuning of the predictions (never of a result): the defaults, or what the environment says, every time a solver is made
        double v[3] = {1e-11, 16e-12, 1.5e-11}, k = 0.25;
        if (const char* e = dfta_knob("LEVELS_NOISE")) sscanf(e, "%lf,%lf,%lf", &v[0], &v[1], &v[2]);   // "rel,abs,secant" of the noise band
        if (const char* e = dfta_knob("LEVELS_SECANT_KAPPA")) k = atof(e);                                // trust in the parabolic correction
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_noise_rel), &v[0], sizeof(double));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_noise_abs), &v[1], sizeof(double));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_secant_noise), &v[2], sizeof(double));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_secant_kappa), &k, sizeof(double));
        const int fp = dfta_knob("LEVELS_NOFIXEDPOINT") ? 0 : 1;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fixed_point), &fp, sizeof(int));
    }
    debug_rounds = dfta_knob("DEBUG_ROUNDS") ? atoi(dfta_knob("DEBUG_ROUNDS")) : 0;
    njobs = static_cast<int>(specs.size());
    if (njobs == 0) return DFTA_OK;
    const int N = g->N;
    // chains: CHAINED -> one chain per potential; BATCHED -> one chain per job.  Both layouts are kept on the
    // device so that a BATCHED solver can run a CHAINED solve (first SCF step, when there are no hints yet).
    std::vector<int> chain_off, chain_off_b, v_off(nV + 1, 0);
    for (int k = 0; k < njobs; ++k) {
        if (specs[k].v < 0 || specs[k].v >= nV || specs[k].l < 0 || specs[k].l > 3) return DFTA_ERR_INVALID;
        if (k > 0 && specs[k].v < specs[k - 1].v) return DFTA_ERR_INVALID;
        v_off[specs[k].v + 1]++;
    }
    for (int v = 0; v < nV; ++v) v_off[v + 1] += v_off[v];
    for (int v = 0; v < nV; ++v) if (v_off[v + 1] > v_off[v]) chain_off.push_back(v_off[v]);
    chain_off.push_back(njobs);
    for (int k = 0; k <= njobs; ++k) chain_off_b.push_back(k);
    nchains_chained = static_cast<int>(chain_off.size()) - 1;
    nchains = (mode == DFTA_LEVELS_CHAINED) ? nchains_chained : njobs;
    // tree depth.  Few jobs (a handful of atoms): the pipelined sweep kernel runs one 64-trial block per compute unit at
    // ~1/4 of the fused kernel's time per point, so the best depth keeps the blocks of a round within ~1.5 passes over
    // the 256 CUs (measured on Rn: depth 10 = 384 blocks beats 9 and 11).  Many jobs: fill the machine with the fused
    // kernel (one wave per SIMD = 65536 lanes).
    int d = tree_depth;
    if (d <= 0) {
        const long active = std::max(1, mode == DFTA_LEVELS_CHAINED ? nchains : njobs);
        d = 6;
        if (active <= 768) { while (d < 14 && ((active << (d + 1)) >> 6) <= 384) ++d; }
        else               { while (d < 14 && (active << (d + 1)) <= 131072L) ++d; }   // up to two waves per SIMD: measured optimum at 960 jobs
    }
    d = std::min(std::max(d, 6), 16);
    depth = d;
    tpj = 1 << d;
    ntrials = static_cast<long>(njobs) * tpj;
    // latency mode (the pipelined kernel's regime: at most ~1.5 passes of one block per compute unit): the slots of a
    // round are re-allotted among the active jobs (k_allot); the budget is one full pass at least
    dynamic = (tree_depth <= 0) && njobs <= 64 && ntrials / 64 <= 384 && dfta_knob("LEVELS_STATIC") == nullptr;
    static_trials = ntrials;
    // A larger job list whose atoms finish one by one (SCF batches: finished atoms are frozen) ends up in that regime too: once at
    // most 64 jobs are live, run() hands the rounds to k_allot through the list of live jobs (room for one pass is kept for it).
    can_switch = (tree_depth <= 0) && !dynamic && mode == DFTA_LEVELS_BATCHED && dfta_knob("LEVELS_STATIC") == nullptr && dfta_knob("LEVELS_NOSWITCH") == nullptr;
    if (dynamic || can_switch) {
        long blocks = std::max(ctx->num_cu, 1);
        if (const char* e = dfta_knob("LEVELS_BUDGET_BLOCKS")) blocks = std::max(64, atoi(e));     // measurements
        budget_trials = 64L * blocks;
        ntrials = std::max<long>(ntrials, budget_trials);
    }
    if (dynamic) budget_trials = ntrials;
    // packed rounds (k_pack): batches whose static layout would have 64-trial trees (up to 192 jobs it has 128 trials per job and
    // the upper half scouts the third bisection: measured on a 12-atom shard, 98 jobs, that is worth 1.6 rounds a step, and the
    // packed layout has no whole blocks of kind ZERO to give to a job that is still counting nodes), unless a depth was asked for
    {
        int min_jobs = 193;
        if (const char* e = dfta_knob("LEVELS_PACK_MIN_JOBS")) min_jobs = atoi(e);              // measurements
        packed = (tree_depth <= 0) && !dynamic && njobs >= min_jobs && dfta_knob("LEVELS_NOPACK") == nullptr;
    }
    // table slots: one per distinct (v, l)
    std::vector<int> slot_v, slot_l;
    std::vector<Job> jobs(njobs);
    {
        std::vector<int> last_of_v(4, -1);          // jobs are ordered by v: only the slots of the current potential can match
        int cur_v = -1;
        for (int k = 0; k < njobs; ++k) {
            if (specs[k].v != cur_v) { cur_v = specs[k].v; std::fill(last_of_v.begin(), last_of_v.end(), -1); }
            int& slot = last_of_v[specs[k].l];
            if (slot < 0) { slot = (int)slot_v.size(); slot_v.push_back(specs[k].v); slot_l.push_back(specs[k].l); }
            Job& j = jobs[k];
            memset(&j, 0, sizeof(Job));
            j.v = specs[k].v; j.n = specs[k].n; j.l = specs[k].l; j.occ = specs[k].occ; j.nodes = specs[k].n - specs[k].l;
            j.slot = slot;
        }
        // sibling: the level of the same potential and l with one node less (its end points bracket this one's, plan_round)
        for (int k = 0, v0 = 0; k < njobs; ++k) {
            if (jobs[k].v != jobs[v0].v) v0 = k;
            jobs[k].sib = -1;
            for (int q = v0; q < njobs && jobs[q].v == jobs[k].v; ++q)
                if (jobs[q].l == jobs[k].l && jobs[q].nodes == jobs[k].nodes - 1) jobs[k].sib = q;
        }
    }
    h_jobs_template = jobs;
    nslots = static_cast<int>(slot_v.size());
    if (packed) {
        pack_dmin = 3; pack_dmax = 12;
        pack_lanes_small = std::max(ctx->num_cu, 64) * 64;   // one pass of one block per compute unit: the pipelined kernel's regime
        pack_dsmall = 3;                                     // ... as long as that leaves every job a tree of this depth
        pack_lanes_large = 131072;                   // two waves per SIMD of the fused kernel
        if (const char* e = dfta_knob("LEVELS_PACK_DMIN")) pack_dmin = std::min(std::max(atoi(e), 1), 8);
        if (const char* e = dfta_knob("LEVELS_PACK_LANES")) pack_lanes_large = std::max(4096, atoi(e));
        if (const char* e = dfta_knob("LEVELS_PACK_LANES_SMALL")) pack_lanes_small = std::max(4096, atoi(e));
        if (const char* e = dfta_knob("LEVELS_PACK_DSMALL")) pack_dsmall = atoi(e);
        depth = pack_dmin;
        tpj = 0;                                     // no fixed share: k_pack lays the trials of a round out
        ntrials = std::max<long>(std::max(pack_lanes_large, pack_lanes_small), static_cast<long>(njobs) * (kPackSpineCap + (1L << pack_dmin))) + 128L * nslots;
        ntrials = (ntrials + 63) & ~63L;
    }
    nwaves = static_cast<int>(ntrials / 64);
    early_match = (dynamic || can_switch) && !g->uniform && dfta_knob("LEVELS_NOEARLYMATCH") == nullptr;

    std::vector<int> wave_slot(nwaves), wave_first(nwaves), wave_cnt(nwaves, 64), wave_job(nwaves);
    for (int w = 0; w < nwaves; ++w) {
        const int q = packed ? njobs : (w * 64) / tpj;
        wave_job[w] = q < njobs ? q : -1;
        wave_slot[w] = q < njobs ? jobs[q].slot : 0;
        wave_first[w] = w * 64;
    }
    h_wave_job = wave_job;
    h_wave_slot = wave_slot;
    tables_dirty = false;

    hipStream_t st = ctx->stream;
#define ALLOC(ptr, type, count) DFTA_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ptr), sizeof(type) * (size_t)(count)))
#define UPLOAD(ptr, vec) DFTA_HIP(ctx, hipMemcpyAsync(ptr, vec.data(), sizeof(vec[0]) * vec.size(), hipMemcpyHostToDevice, st))
    ALLOC(d_jobs, Job, njobs);
    ALLOC(d_chain_off, int, chain_off.size()); UPLOAD(d_chain_off, chain_off);
    ALLOC(d_chain_off_b, int, chain_off_b.size()); UPLOAD(d_chain_off_b, chain_off_b);
    ALLOC(d_v_off, int, v_off.size()); UPLOAD(d_v_off, v_off);
    ALLOC(d_slot_v, int, nslots); UPLOAD(d_slot_v, slot_v);
    ALLOC(d_slot_l, int, nslots); UPLOAD(d_slot_l, slot_l);
    ALLOC(d_tab, double2, (size_t)nslots * N);
    ALLOC(d_E, double, ntrials); ALLOC(d_limit, int, ntrials); ALLOC(d_start, int, ntrials);
    ALLOC(d_us, double, ntrials); ALLOC(d_us1, double, ntrials); ALLOC(d_count, int, ntrials); ALLOC(d_u0, double, ntrials); ALLOC(d_phi, double, ntrials); ALLOC(d_istop, int, ntrials); ALLOC(d_trip, int, ntrials);
    ALLOC(d_wave_kind, int, nwaves);
    ALLOC(d_wave_slot, int, nwaves); UPLOAD(d_wave_slot, wave_slot);
    ALLOC(d_wave_first, int, nwaves); UPLOAD(d_wave_first, wave_first);
    ALLOC(d_wave_cnt, int, nwaves); UPLOAD(d_wave_cnt, wave_cnt);
    ALLOC(d_wave_job, int, nwaves); UPLOAD(d_wave_job, wave_job);
    if (packed) {
        std::vector<int> slot_off(nslots + 1, 0), slot_jobs(njobs);
        for (int k = 0; k < njobs; ++k) slot_off[jobs[k].slot + 1]++;
        for (int q = 0; q < nslots; ++q) slot_off[q + 1] += slot_off[q];
        std::vector<int> fill(slot_off.begin(), slot_off.end() - 1);
        for (int k = 0; k < njobs; ++k) slot_jobs[fill[jobs[k].slot]++] = k;
        ALLOC(d_slot_off, int, nslots + 1); UPLOAD(d_slot_off, slot_off);
        ALLOC(d_slot_jobs, int, njobs); UPLOAD(d_slot_jobs, slot_jobs);
        ALLOC(d_gsz, int, 2 * nslots); ALLOC(d_goff, int, 2 * nslots);
        ALLOC(d_lane_job, int, ntrials);
        ALLOC(d_pack_out, int, 4);
        DFTA_HIP(ctx, hipStreamSynchronize(st));     // the vectors above are the sources of the copies
    }
    ALLOC(d_counters, unsigned long long, 4);
    if (can_switch) ALLOC(d_live, int, 64);
    ALLOC(d_Psi, double, (size_t)njobs * N);
    ALLOC(d_Q, double, (size_t)njobs * N);
    ALLOC(d_jE, double, njobs); ALLOC(d_jslot, int, njobs); ALLOC(d_jl, int, njobs); ALLOC(d_jstart, int, njobs);
    ALLOC(d_jmatched, int, njobs); ALLOC(d_jstart_keep, int, njobs);
    ALLOC(d_snapE, double, njobs); ALLOC(d_snapReady, int, njobs); ALLOC(d_jtake, int, njobs);
    if (early_match) {
        DFTA_HIP(ctx, hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
        DFTA_HIP(ctx, hipEventCreateWithFlags(&ev_walk, hipEventDisableTiming));
        DFTA_HIP(ctx, hipEventCreateWithFlags(&ev_early, hipEventDisableTiming));
        DFTA_HIP(ctx, hipEventCreateWithFlags(&ev_taken, hipEventDisableTiming));
    }
    ALLOC(d_jus, double, njobs); ALLOC(d_jus1, double, njobs); ALLOC(d_jmp, int, njobs);
    ALLOC(d_slot_min, double, nslots);
    ALLOC(d_bounds, double2, (size_t)nslots * dfta_bounds_stride(g));
    if (sweep_mode == DFTA_SWEEPS_TOLERANCE && dfta_scan_supported(g)) {      // scan.hip: interleaved tables + per-lane {min, max}
        const int trc = dfta_scan_tables_create(ctx, g, nslots, &scan_tb);
        if (trc) return trc;
        ALLOC(d_scan_live, int, njobs);
        ALLOC(d_scan_xch, unsigned long long, 32 * (size_t)njobs);
    }
#undef ALLOC
#undef UPLOAD
    DFTA_HIP(ctx, hipEventCreate(&ev[0]));
    DFTA_HIP(ctx, hipEventCreate(&ev[1]));
    DFTA_HIP(ctx, hipMemsetAsync(d_count, 0, sizeof(int) * ntrials, st));
    DFTA_HIP(ctx, hipMemsetAsync(d_u0, 0, sizeof(double) * ntrials, st));
    DFTA_HIP(ctx, hipMemsetAsync(d_trip, 0, sizeof(int) * ntrials, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    return DFTA_OK;
}

// Solve all levels for the potentials dV (device, nV*N).
//   job_bottom (host, njobs): BottomEnergy at entry of LocateInterval for every job.  CHAINED uses only the entry
//   of the first job of each potential (-Z^2-1, DFTAtom.cpp:407) and hands E-3 from level to level (DFTAtom.cpp:541);
//   BATCHED starts every level from its own entry (the caller's hint: E_{k-1} of the previous SCF step - 3).
int LevelSolver::run(const double* dV, const double* job_bottom, int run_mode, double* dNewDensity, LevelStats* stats,
                     const unsigned char* frozen)
{
    if (njobs == 0) return DFTA_OK;
    const int N = g->N;
    hipStream_t st = ctx->stream;
    const bool chained = (run_mode == DFTA_LEVELS_CHAINED);
    const int* d_chains = chained ? d_chain_off : d_chain_off_b;
    const int run_chains = chained ? nchains_chained : njobs;
    std::vector<Job> jobs = h_jobs_template;
    int nfrozen = 0;
    // this run's layout: the solver's own, or -- a batch most of whose atoms have finished -- latency mode over the live jobs
    std::vector<int> live;
    if (frozen && h_last.size() == jobs.size())
        for (int k = 0; k < njobs; ++k) if (!frozen[k]) live.push_back(k);
    const bool sw = can_switch && !chained && frozen && h_last.size() == jobs.size() && !live.empty() && live.size() <= 64;
    const bool dyn = dynamic || sw;
    const bool pk = packed && !sw;
    const bool early = early_match && dyn;
    if (sw) {
        DFTA_HIP(ctx, hipMemcpyAsync(d_live, live.data(), sizeof(int) * live.size(), hipMemcpyHostToDevice, st));
        tables_dirty = true;
    } else if (tables_dirty) {           // back from latency mode: the solver's own block tables
        DFTA_HIP(ctx, hipMemcpyAsync(d_wave_job, h_wave_job.data(), sizeof(int) * h_wave_job.size(), hipMemcpyHostToDevice, st));
        DFTA_HIP(ctx, hipMemcpyAsync(d_wave_slot, h_wave_slot.data(), sizeof(int) * h_wave_slot.size(), hipMemcpyHostToDevice, st));
        tables_dirty = false;
    }
    for (int k = 0; k < njobs; ++k) {
        Job& j = jobs[k];
        if (frozen && frozen[k] && h_last.size() == jobs.size()) {
            j = h_last[k];             // E, interval, convergence flag, match point, history: as the last solve left them
            j.phase = PH_DONE;
            j.frozen = 1;
            j.spine = 0;
            j.capz = 0;
            ++nfrozen;
            continue;
        }
        j.frozen = 0;
        j.tbase = (pk || sw) ? 0 : k * tpj;      // packed rounds: k_pack lays the trials out (latency mode: k_allot)
        j.tcap = (pk || sw) ? 0 : tpj;
        j.bottom0 = job_bottom[k];
        const bool first = (k == 0 || jobs[k].v != jobs[k - 1].v);
        if (!chained || first) { j.phase = PH_TOP; j.toe = 50; j.boe = j.bottom0; }   // DFTAtom.cpp:499
        else j.phase = PH_WAIT;
        // path prediction from the previous solve of the same job list (speculation only)
        if (use_prediction && h_last.size() == jobs.size()) {
            for (int ph = 0; ph < 3; ++ph) {
                j.pred_bits[ph] = h_last[k].cur_bits[ph];
                j.pred_len[ph] = h_last[k].cur_len[ph];
                j.trust[ph] = h_last[k].trust[ph];
                // the very first comparison has no history: trust what a one-step-old prediction typically gives
                // (no spine from it: a miss forfeits the whole tree of a round, three predicted bits are not worth that)
                if (h_last[k].pred_len[ph] == 0) j.trust[ph] = 0;
            }
            const Job& h = h_last[k];
            const double T[3] = {h.top, h.bottom, h.E};
            for (int ph = 0; ph < 3; ++ph) {
                j.hist_T[ph] = T[ph];
                j.hist_d[ph] = h.hist_ok >= 1 ? fabs(T[ph] - h.hist_T[ph]) : -1.0;
            }
            j.hist_ok = h.hist_ok >= 1 ? 2 : 1;
        } else {
            j.hist_ok = 0;
        }
        j.miss = 0;
        j.capz = 0;
        j.se_state = 0;
        j.se_stop = 0;
        j.se_sl = 0;
        j.se_tok = 0;
        j.se_plo = j.se_phi = NAN;
        j.sc_is[0] = j.sc_is[1] = j.sc_is[2] = -1;
        j.sc_ok = 0;
        j.use_sp = 0;
        j.sp_len = 0;
        j.sp_bits = 0;
        j.spine = 0;           // planned on the device (k_plan below), after the bottoms have been clamped
        j.phase_done = 0;
    }
    DFTA_HIP(ctx, hipMemcpyAsync(d_jobs, jobs.data(), sizeof(Job) * njobs, hipMemcpyHostToDevice, st));
    DFTA_HIP(ctx, hipMemsetAsync(d_counters, 0, sizeof(unsigned long long) * 4, st));
    DFTA_HIP(ctx, hipMemsetAsync(d_jmatched, 0, sizeof(int) * njobs, st));
    if (nfrozen == njobs) {            // nothing to solve: Psi, density contributions and job records stand
        DFTA_HIP(ctx, hipStreamSynchronize(st));      // `jobs` is the source of the copy above
        if (stats) *stats = LevelStats();
        return DFTA_OK;
    }
    int rc = dfta_launch_build_tab(ctx, g, d_tab, dV, d_slot_v, d_slot_l, nslots, d_bounds);
    if (rc) return rc;
    if (!chained && clamp_bottoms) {
        hipLaunchKernelGGL(k_slot_min, dim3(nslots), dim3(1024), 0, st, d_tab, N, d_slot_min);
        DFTA_CHECK_LAUNCH(ctx);
        hipLaunchKernelGGL(k_clamp_bottoms, dim3((njobs + 63) / 64), dim3(64), 0, st, d_jobs, njobs, d_slot_min);
        DFTA_CHECK_LAUNCH(ctx);
    }

    auto plan = [&]() -> int {          // spines (and, in latency mode, the slots) of the next round
        if (dyn)
            hipLaunchKernelGGL(k_allot, dim3(1), dim3(64), 0, st, d_jobs, sw ? (int)live.size() : njobs, (int)budget_trials, use_prediction ? 0 : 1, d_wave_job,
                               d_wave_slot, sw ? d_live : nullptr);
        else
            hipLaunchKernelGGL(k_plan, dim3((njobs + 63) / 64), dim3(64), 0, st, d_jobs, njobs, use_prediction ? 0 : 1, pk ? (1 << 14) : 0);
        DFTA_CHECK_LAUNCH(ctx);
        if (pk) {
            hipLaunchKernelGGL(k_pack, dim3(1), dim3(kPackThreads), 0, st, d_jobs, njobs, nslots, d_slot_off, d_slot_jobs, pack_lanes_small,
                               pack_dsmall, pack_lanes_large, pack_dmin, pack_dmax, d_gsz, d_goff, d_lane_job, d_wave_slot, d_pack_out);
            DFTA_CHECK_LAUNCH(ctx);
            hipLaunchKernelGGL(k_pack_lanes, dim3((njobs + 3) / 4), dim3(256), 0, st, d_jobs, njobs, d_lane_job);
            DFTA_CHECK_LAUNCH(ctx);
        }
        return DFTA_OK;
    };
    // TOLERANCE MODE of the sweeps (scan.hip, opt-in): every level's three bisections by one workgroup, start to end on the device --
    // no rounds, no speculation.  A sweep the scan cannot decide (non-finite values, f >= 12 in a step row: never seen for the
    // potential of an SCF) sends the whole solve to the exact kernels below.
    bool scan = sweep_mode == DFTA_SWEEPS_TOLERANCE && scan_tb.tabv != nullptr;
    // the match solve and the normalisation in the same kernel (SCAN_NOMATCH: the exact k_match / k_normalize follow, measurements)
    const int scan_match_mode = dfta_knob("SCAN_NOMATCH") ? 0 : (integ_rule == DFTA_INT_SIMPSON38 ? 2 : 1);
    float ms_scan = 0;
    if (scan) {
        rc = dfta_launch_scan_build_tab(ctx, g, scan_tb, dV, d_slot_v, d_slot_l);
        if (rc) return rc;
        // a handful of levels leave compute units idle: K = 15, 7 or 3 workgroups per level take a depth-4, 3 or 2 bisection tree per round
        std::vector<int> live_jobs;
        for (int k = 0; k < njobs; ++k) if (!jobs[k].frozen) live_jobs.push_back(k);
        int K = 1;
        if (!chained && !live_jobs.empty()) {
            for (int cand : {15, 7, 3}) if ((long)live_jobs.size() * cand <= ctx->num_cu) { K = cand; break; }
            if (const char* e = dfta_knob("SCAN_GROUP")) { const int f = atoi(e); K = (f == 15 || f == 7 || f == 3) && (long)live_jobs.size() * f <= ctx->num_cu ? f : 1; }
        }
        DFTA_HIP(ctx, hipEventRecord(ev[0], st));
        rc = DFTA_ERR_NOT_CONVERGED;
        if (K > 1) {
            DFTA_HIP(ctx, hipMemcpyAsync(d_scan_live, live_jobs.data(), sizeof(int) * live_jobs.size(), hipMemcpyHostToDevice, st));
            DFTA_HIP(ctx, hipMemsetAsync(d_jstart_keep, 0xff, sizeof(int) * njobs, st));      // -1: frozen (the live jobs write their cut-off index)
            rc = dfta_launch_scan_levels_group(ctx, g, d_jobs, d_scan_live, (int)live_jobs.size(), K, scan_tb, dfta_knob("LEVELS_NOFIXEDPOINT") ? 0 : 1, d_counters,
                                               d_scan_xch, scan_match_mode, d_Psi, d_jstart_keep);
            if (rc && rc != DFTA_ERR_NOT_CONVERGED) return rc;
        }
        if (rc == DFTA_ERR_NOT_CONVERGED) {
            K = 1;
            rc = dfta_launch_scan_levels(ctx, g, d_jobs, d_chains, run_chains, chained ? 1 : 0, scan_tb, dfta_knob("LEVELS_NOFIXEDPOINT") ? 0 : 1, d_counters,
                                         scan_match_mode, d_Psi, d_jstart_keep);
        }
        if (rc) return rc;
        scan_group = K;
        DFTA_HIP(ctx, hipEventRecord(ev[1], st));
        unsigned long long flag = 0;
        DFTA_HIP(ctx, hipMemcpyAsync(&flag, d_counters + 3, sizeof(flag), hipMemcpyDeviceToHost, st));
        DFTA_HIP(ctx, hipStreamSynchronize(st));
        if (stats) DFTA_HIP(ctx, hipEventElapsedTime(&ms_scan, ev[0], ev[1]));
        if (flag) {                  // back to the exact kernels with the job records as they were uploaded (a trial the scan could not decide, or a lost group member)
            scan = false;
            ++scan_fallbacks;
            DFTA_HIP(ctx, hipMemcpyAsync(d_jobs, jobs.data(), sizeof(Job) * njobs, hipMemcpyHostToDevice, st));
            DFTA_HIP(ctx, hipMemsetAsync(d_counters, 0, sizeof(unsigned long long) * 4, st));
            if (!chained && clamp_bottoms) {
                hipLaunchKernelGGL(k_clamp_bottoms, dim3((njobs + 63) / 64), dim3(64), 0, st, d_jobs, njobs, d_slot_min);
                DFTA_CHECK_LAUNCH(ctx);
            }
        }
    }
    if (!scan) {
        rc = plan();
        if (rc) return rc;
    }
    // trials of the coming round: the whole static / latency-mode layout, or what k_pack has just laid out
    long round_trials = dyn ? budget_trials : static_trials;
    int pack_out[4] = {0, 0, 0, 0};
    if (pk && !scan) {
        DFTA_HIP(ctx, hipMemcpyAsync(pack_out, d_pack_out, sizeof(pack_out), hipMemcpyDeviceToHost, st));
        DFTA_HIP(ctx, hipStreamSynchronize(st));
        round_trials = pack_out[0];
    }
    if (early && !scan) {
        hipLaunchKernelGGL(k_job_slots, dim3((njobs + 63) / 64), dim3(64), 0, st, d_jobs, njobs, d_jslot, d_jl);
        DFTA_CHECK_LAUNCH(ctx);
        DFTA_HIP(ctx, hipMemsetAsync(d_jE, 0xff, sizeof(double) * njobs, st));      // NaN: "no energy yet" (k_take_ready)
    }
    int* d_ndone = reinterpret_cast<int*>(d_counters + 2);
    int rounds = scan ? 1 : 0;
    float ms_sweep = ms_scan;
    const int max_rounds = 4096;
    int done_seen = nfrozen;
    bool early_pending = false;
    while (!scan && rounds < max_rounds) {
        dfta_range r_round("dfta: level-search round (expand, sweeps, scout, walk, plan)");
        if (round_trials <= 0 || round_trials > ntrials) { snprintf(ctx->err, sizeof(ctx->err), "level solver: packed round of %ld trials (room for %ld)", round_trials, ntrials); return DFTA_ERR_HIP; }
        const int round_waves = static_cast<int>(round_trials / 64);
        hipLaunchKernelGGL(k_expand, dim3((unsigned)((round_trials + 255) / 256)), dim3(256), 0, st, d_jobs, pk ? d_lane_job : d_wave_job, pk ? 0 : 6,
                           (int)round_trials, g->d_r, N, g->delta, g->far_arg_threshold, d_E, d_limit, d_start, d_us, d_us1, d_wave_kind, d_counters, g->uniform,
                           g->Rmax, g->h);
        DFTA_CHECK_LAUNCH(ctx);
        if (stats) DFTA_HIP(ctx, hipEventRecord(ev[0], st));
        rc = dfta_launch_sweep(ctx, g, DFTA_SWEEP_COUNT, d_wave_kind, round_waves, d_tab, d_wave_slot, d_wave_first, d_wave_cnt, d_E,
                               d_limit, d_start, d_us, d_us1, d_count, d_u0, stats ? d_trip : nullptr, stats ? d_counters + 1 : nullptr, g->uniform ? nullptr : d_bounds, d_phi,
                               d_istop, d_slot_l);
        if (rc) return rc;
        if (stats) DFTA_HIP(ctx, hipEventRecord(ev[1], st));
        if (!pk) {          // packed rounds have no scouts (capz == tcap)
            hipLaunchKernelGGL(k_scout, dim3(njobs), dim3(64), 0, st, d_jobs, d_E, d_start, d_u0);
            DFTA_CHECK_LAUNCH(ctx);
        }
        DFTA_HIP(ctx, hipMemsetAsync(d_ndone, 0, sizeof(int), st));
        hipLaunchKernelGGL(k_walk, dim3((run_chains + 63) / 64), dim3(64), 0, st, d_jobs, d_chains, run_chains, d_count, d_u0, d_phi, d_istop, stats ? d_trip : nullptr, d_tab, N, d_ndone);
        DFTA_CHECK_LAUNCH(ctx);
        if (early) {
            DFTA_HIP(ctx, hipStreamWaitEvent(st, ev_taken, 0));      // the second stream has consumed the previous snapshot (no-op if none was taken)
            hipLaunchKernelGGL(k_snapshot_done, dim3((njobs + 63) / 64), dim3(64), 0, st, d_jobs, njobs, d_snapE, d_snapReady);
            DFTA_CHECK_LAUNCH(ctx);
            DFTA_HIP(ctx, hipEventRecord(ev_walk, st));
        }
        rc = plan();
        if (rc) return rc;
        int ndone = 0;
        DFTA_HIP(ctx, hipMemcpyAsync(&ndone, d_ndone, sizeof(int), hipMemcpyDeviceToHost, st));
        if (pk) DFTA_HIP(ctx, hipMemcpyAsync(pack_out, d_pack_out, sizeof(pack_out), hipMemcpyDeviceToHost, st));
        DFTA_HIP(ctx, hipStreamSynchronize(st));
        const long this_round = round_trials;
        if (pk) round_trials = pack_out[0];
        if (early && ndone > done_seen && ndone < njobs) {
            // some levels have their eigenvalue while others still search: their match solves start now, on the second stream,
            // under the next round's sweeps (two waves and 8 KB of LDS per level fit next to a sweep block)
            done_seen = ndone;
            DFTA_HIP(ctx, hipStreamWaitEvent(st2, ev_walk, 0));
            hipLaunchKernelGGL(k_take_ready, dim3((njobs + 63) / 64), dim3(64), 0, st2, d_snapE, d_snapReady, d_jmatched, njobs, d_jE, d_jtake);
            DFTA_CHECK_LAUNCH(ctx);
            DFTA_HIP(ctx, hipEventRecord(ev_taken, st2));
            int erc = dfta_launch_boundary(ctx, g, d_jE, njobs, d_jstart, d_jus, d_jus1, 1, d_jl, d_Q, st2);
            if (!erc) {
                hipLaunchKernelGGL(k_mask_ready, dim3((njobs + 63) / 64), dim3(64), 0, st2, d_jtake, njobs, d_jstart, d_jmatched, d_jstart_keep);
                DFTA_CHECK_LAUNCH(ctx);
                erc = dfta_launch_match(ctx, g, njobs, d_tab, d_jslot, d_jE, d_jstart, d_jus, d_jus1, d_jl, d_Psi, d_Q, d_jmp, d_bounds, d_Q, st2);
            }
            if (erc) return erc;
            DFTA_HIP(ctx, hipEventRecord(ev_early, st2));
            early_pending = true;
        }
        if (stats) {
            float ms = 0;
            DFTA_HIP(ctx, hipEventElapsedTime(&ms, ev[0], ev[1]));
            ms_sweep += ms;
            if (debug_rounds && pk)
                fprintf(stderr, "   packed round %d: %ld trials in %.3f ms; next: %d trials, depth %d, %d jobs searching\n", rounds + 1, this_round, ms, pack_out[0], pack_out[1], pack_out[2]);
        }
        ++rounds;
        if (debug_rounds) {      // per-round census of the jobs (phase/decisions taken), stderr
            std::vector<Job> dbg(njobs);
            DFTA_HIP(ctx, hipMemcpy(dbg.data(), d_jobs, sizeof(Job) * njobs, hipMemcpyDeviceToHost));
            fprintf(stderr, "round %2d |", rounds);
            for (int q = 0; q < njobs; ++q)
                if (dbg[q].phase != PH_DONE) fprintf(stderr, " %d:%d/%d", q, dbg[q].phase, dbg[q].phase_done);
            fprintf(stderr, "\n");
            if (debug_rounds >= 3)
                for (int q = 0; q < njobs; ++q) {
                    const Job& J = dbg[q];
                    fprintf(stderr, "   J %2d ph %d done %d l %d boe %.17g toe %.17g top %.17g sc_ok %d sc_lo %.17g sc_hi %.17g spine %d use_sp %d miss %d tcap %d\n",
                            q, J.phase, J.phase_done, J.l, J.boe, J.toe, J.top, J.sc_ok, J.sc_lo, J.sc_hi, J.spine, J.use_sp, J.miss, J.tcap);
                }
            if (debug_rounds >= 2)
                for (int q = 0; q < njobs; ++q) {
                    const Job& J = dbg[q];
                    if (J.phase != PH_TOP) continue;
                    fprintf(stderr, "   job %2d l=%d w=%.3e ok=%d spine=%d use_sp=%d miss=%d | a: is=%d phi=%.4e  b: is=%s%d phi=%.4e  c: de=%.3e is=%s%d phi=%.4e | pred w/e=%.3e\n",
                            q, J.l, J.toe - J.boe, J.sc_ok, J.spine, J.use_sp, J.miss, J.sc_is[0], J.sc_phi[0],
                            (J.sc_is[1] >= 0 && (J.sc_is[1] & kStopOver)) ? "o" : "", J.sc_is[1] < 0 ? -1 : (J.sc_is[1] & ~kStopOver), J.sc_phi[1],
                            J.sc_e[2] - J.sc_e[0], (J.sc_is[2] >= 0 && (J.sc_is[2] & kStopOver)) ? "o" : "", J.sc_is[2] < 0 ? -1 : (J.sc_is[2] & ~kStopOver), J.sc_phi[2],
                            J.sc_ok ? (J.toe - J.boe) / (J.sc_hi - J.sc_lo) : 0.0);
                }
        }
        if (ndone >= njobs) break;
    }
    if (rounds >= max_rounds) { snprintf(ctx->err, sizeof(ctx->err), "level solver did not terminate"); return DFTA_ERR_NOT_CONVERGED; }

    // wavefunctions: match (the levels that were not matched while the others searched), normalise, accumulate
    if (scan && scan_match_mode) {
        // k_scan_levels has matched (and, with Simpson 3/8, normalised) every live level
        if (scan_match_mode == 1) {
            hipLaunchKernelGGL(k_normalize, dim3(njobs), dim3(kNormThreads), 0, st, d_Psi, d_Q, N, g->d_eh, g->d_cnst, d_jstart_keep, g->uniform ? g->h : 1.0, integ_rule);
            DFTA_CHECK_LAUNCH(ctx);
        }
    } else {
        if (early_pending) DFTA_HIP(ctx, hipStreamWaitEvent(st, ev_early, 0));       // the early solves use the same per-job scratch arrays
        hipLaunchKernelGGL(k_job_energies, dim3((njobs + 63) / 64), dim3(64), 0, st, d_jobs, njobs, d_jE, d_jslot, d_jl);
        DFTA_CHECK_LAUNCH(ctx);
        rc = dfta_launch_boundary(ctx, g, d_jE, njobs, d_jstart, d_jus, d_jus1, 1, d_jl, d_Q /* uniform: start value at the first node, one per job */);
        if (rc) return rc;
        // cut-off index -1 = skipped by k_match: frozen jobs (the result of their last solve stands) and jobs matched already;
        // d_jstart_keep: the cut-off index of every job that was matched in this run (-1: frozen)
        hipLaunchKernelGGL(k_mask_rest, dim3((njobs + 63) / 64), dim3(64), 0, st, d_jobs, njobs, d_jstart, d_jmatched, d_jstart_keep);
        DFTA_CHECK_LAUNCH(ctx);
        rc = dfta_launch_match(ctx, g, njobs, d_tab, d_jslot, d_jE, d_jstart, d_jus, d_jus1, d_jl, d_Psi, d_Q, d_jmp, g->uniform ? nullptr : d_bounds,
                               d_Q);
        if (rc) return rc;
        hipLaunchKernelGGL(k_store_match, dim3((njobs + 63) / 64), dim3(64), 0, st, d_jobs, njobs, d_jmp, d_jstart_keep);
        DFTA_CHECK_LAUNCH(ctx);
        hipLaunchKernelGGL(k_normalize, dim3(njobs), dim3(kNormThreads), 0, st, d_Psi, d_Q, N, g->d_eh, g->d_cnst, nfrozen ? d_jstart_keep : nullptr, g->uniform ? g->h : 1.0, integ_rule);
        DFTA_CHECK_LAUNCH(ctx);
    }
    if (dNewDensity) {
        hipLaunchKernelGGL(k_accumulate_density, dim3(std::min(256, (N + 255) / 256), nV), dim3(256), 0, st, d_Psi, d_jobs, d_v_off,
                           N, dNewDensity);
        DFTA_CHECK_LAUNCH(ctx);
    }
    if (stats) {
        unsigned long long cnt[2];
        DFTA_HIP(ctx, hipMemcpyAsync(cnt, d_counters, sizeof(cnt), hipMemcpyDeviceToHost, st));
        DFTA_HIP(ctx, hipStreamSynchronize(st));
        stats->rounds = rounds;
        stats->sweeps_issued = static_cast<long>(cnt[0]) + 2L * (njobs - nfrozen);    // + inward/outward halves of the match solve
        stats->points_traversed = static_cast<long>(cnt[1]);
        stats->ms_sweep = ms_sweep;
        stats->layout = scan ? 4 : (sw ? 3 : (dynamic ? 1 : (pk ? 2 : 0)));
    }
    return DFTA_OK;
}

int LevelSolver::fetch_jobs(std::vector<Job>& out)
{
    out.resize(njobs);
    if (njobs == 0) return DFTA_OK;
    DFTA_HIP(ctx, hipMemcpyAsync(out.data(), d_jobs, sizeof(Job) * njobs, hipMemcpyDeviceToHost, ctx->stream));
    DFTA_HIP(ctx, hipStreamSynchronize(ctx->stream));
    h_last = out;            // decision strings of this solve predict the paths of the next one
    return DFTA_OK;
}

}  // namespace dfta

extern "C" int dfta_solve_levels(dfta_ctx* ctx, const dfta_grid* g, int mode, int tree_depth, int nV, const double* V,
                                 const double* bottom0, const double* bottom_hint, int nlevels, const int* vidx, const int* n,
                                 const int* l, const int* occ,
                                 dfta_level_result* results, double* newDensity, double* Eelectronic, double* Psi_out,
                                 long* issued_sweeps)
{
    if (!ctx || !g) return DFTA_ERR_INVALID;
    DFTA_ENTER(ctx);
    DFTA_REQUIRE(ctx, V && bottom0 && n && l && occ && results && nV > 0 && nlevels > 0, "null input");
    const bool scan_sweeps = (mode & DFTA_LEVELS_SCAN_SWEEPS) != 0;
    mode &= ~DFTA_LEVELS_SCAN_SWEEPS;
    DFTA_REQUIRE(ctx, mode == DFTA_LEVELS_CHAINED || mode == DFTA_LEVELS_BATCHED, "mode");
    DFTA_REQUIRE(ctx, !scan_sweeps || dfta_scan_supported(g), "the tolerance mode of the sweeps needs a logarithmic grid of 12 .. 20 multigrid levels");
    const int N = g->N;
    std::vector<dfta::JobSpec> specs(nlevels);
    for (int k = 0; k < nlevels; ++k) specs[k] = {vidx ? vidx[k] : 0, n[k], l[k], occ[k]};
    dfta::LevelSolver solver;
    solver.sweep_mode = scan_sweeps ? DFTA_SWEEPS_TOLERANCE : DFTA_SWEEPS_EXACT;
    int rc = solver.setup(ctx, g, mode, tree_depth, nV, specs);
    if (rc) { if (rc == DFTA_ERR_INVALID) snprintf(ctx->err, sizeof(ctx->err), "levels must be grouped by potential, l in 0..3"); return rc; }
    hipStream_t st = ctx->stream;
    DevBuf<double> dV, dND;
    DFTA_HIP(ctx, dV.alloc((size_t)nV * N));
    DFTA_HIP(ctx, hipMemcpyAsync(dV.p, V, sizeof(double) * (size_t)nV * N, hipMemcpyHostToDevice, st));
    if (newDensity) {
        DFTA_HIP(ctx, dND.alloc((size_t)nV * N));
        DFTA_HIP(ctx, hipMemcpyAsync(dND.p, newDensity, sizeof(double) * (size_t)nV * N, hipMemcpyHostToDevice, st));
    }
    dfta::LevelStats stats;
    std::vector<double> job_bottom(nlevels);
    for (int k = 0; k < nlevels; ++k) job_bottom[k] = bottom_hint ? bottom_hint[k] : bottom0[specs[k].v];
    solver.clamp_bottoms = (bottom_hint == nullptr);    // explicit hints are taken as given
    rc = solver.run(dV.p, job_bottom.data(), mode, dND.p, &stats);
    if (rc) return rc;
    std::vector<dfta::Job> jobs;
    rc = solver.fetch_jobs(jobs);
    if (rc) return rc;
    if (Eelectronic) for (int v = 0; v < nV; ++v) Eelectronic[v] = 0;
    bool allconv = true;
    for (int k = 0; k < nlevels; ++k) {
        const dfta::Job& j = jobs[k];
        results[k].E = j.E; results[k].top = j.top; results[k].bottom = j.bottom; results[k].n_count = j.n_count;
        results[k].n_zero = j.n_zero; results[k].converged = j.converged; results[k].matchPoint = j.matchPoint;
        results[k].status = j.status;
        if (Eelectronic) Eelectronic[j.v] += j.occ * j.E;            // DFTAtom.cpp:561
        allconv = allconv && j.converged;
    }
    if (newDensity) DFTA_HIP(ctx, hipMemcpyAsync(newDensity, dND.p, sizeof(double) * (size_t)nV * N, hipMemcpyDeviceToHost, st));
    if (Psi_out) DFTA_HIP(ctx, hipMemcpyAsync(Psi_out, solver.d_Psi, sizeof(double) * (size_t)nlevels * N, hipMemcpyDeviceToHost, st));
    DFTA_HIP(ctx, hipStreamSynchronize(st));
    if (issued_sweeps) *issued_sweeps = stats.sweeps_issued;
    (void)allconv;
    return DFTA_OK;
}
