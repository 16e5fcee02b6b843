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

#include "levels_device.inc"

// ---- expand: trial energies of the current round of every job, plus their far boundary values ----------------------
// qctl / qlist (may be null): the round's work queue for the balanced launch of the fused sweeps (numerov.hip:k_sweep_queue).  Every
// 64-trial block with an active trial is entered into one of kSweepQueueClasses lists by its expected length -- the cut-off index of
// its outermost trial, halved where CountNodes will stop at an inner turning point (l > 0) -- so that the launch's waves take the
// longest blocks first; blocks without an active trial are not entered at all.
__global__ __launch_bounds__(256) void k_expand(const dfta::Job* __restrict__ jobs, const int* __restrict__ wave_job, int wshift, int ntrials, const double* __restrict__ r,
                                                int N, double delta, double far_thr, double* __restrict__ E,
                                                int* __restrict__ limit, int* __restrict__ start, double* __restrict__ us,
                                                double* __restrict__ us1, int* __restrict__ wave_kind,
                                                unsigned long long* __restrict__ issued, int uniform, double Rmax, double hstep,
                                                int* __restrict__ qctl, int* __restrict__ qlist, int qcap)
{
    const int gt = blockIdx.x * blockDim.x + threadIdx.x;
    if (gt >= ntrials) return;       // ntrials is a multiple of 64: whole waves leave
    const int job = wave_job[gt >> wshift];   // one entry per 64-trial block (wshift 6) or per trial (packed rounds: wshift 0)
    bool active = false, zero_kind = true;
    int st = 0, key = 0;
    if (job < 0) {                    // slots that no job owns this round
        E[gt] = 0; limit[gt] = 0; start[gt] = 0;
    } else {
        const dfta::Job j = jobs[job];
        const TrialSpec ts = trial_spec<false>(j, jobs, gt - j.tbase);
        const double e = ts.e;
        active = ts.active;
        zero_kind = ts.zero_kind;
        E[gt] = e;
        limit[gt] = j.nodes;
        if (active && uniform) {
            // uniform grid (Numerov.h:32-35,43-56,274-296): start at min(Rmax, 200 / sqrt(2|E|)), index (long)(startPoint / h)
            const double s = sqrt(2. * fabs(e));
            const double mr = 200. / s;
            const double sp = mr < Rmax ? mr : Rmax;
            st = static_cast<int>(static_cast<long>(sp / hstep));
            us[gt] = exp(-sp * s);
            us1[gt] = exp(-(sp - hstep) * s);
        } else if (active) {
            double a, b;
            st = trial_boundary(e, r, N, delta, far_thr, a, b);
            us[gt] = a;
            us1[gt] = b;
        }
        start[gt] = st;
        key = active ? ((zero_kind || j.l == 0) ? st : st / 2) : 0;
    }
    // the block's kind: its first trial's (a block of the static / latency layouts has one owner, a packed block one (slot, kind) group)
    const int kind0 = __shfl(zero_kind ? 1 : 0, 0);
    if ((gt & 63) == 0) wave_kind[gt >> 6] = kind0 ? DFTA_SWEEP_ZERO : DFTA_SWEEP_COUNT;
    const unsigned long long m = __ballot(active);
    if (issued && (threadIdx.x & 63) == 0 && m) atomicAdd(issued, (unsigned long long)__popcll(m));
    if (qctl) {
        for (int off = 32; off > 0; off >>= 1) key = max(key, __shfl_xor(key, off));
        if ((gt & 63) == 0 && m) {
            int cls = kSweepQueueClasses - 1 - static_cast<int>((static_cast<long long>(key) * kSweepQueueClasses) / N);
            cls = cls < 0 ? 0 : (cls >= kSweepQueueClasses ? kSweepQueueClasses - 1 : cls);
            const int pos = atomicAdd(&qctl[cls], 1);
            qlist[(size_t)cls * qcap + pos] = gt >> 6;
        }
    }
}

__global__ __launch_bounds__(64) void k_scout(dfta::Job* __restrict__ jobs, const double* __restrict__ E,
                                              const int* __restrict__ start, const double* __restrict__ u0)
{
    scout_job(jobs + blockIdx.x, threadIdx.x, E, start, u0);
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

// TopEnergy of every live level as the scan search found it (scan.hip, on a copy of the records) becomes the centre of the bracket from
// which plan_round lays the first spine of the exact search's FIRST bisection -- speculation only.  Half width = `factor` (1.5) x the
// scan's asserted distance from the exact path, 6e-11 |T| + 6e-10 Ha (tests/test_gpu_scan.py), less the 1e-10 |T| that plan_round adds
// for brackets from SCF histories.  Measured (Rn, steps 5 - 24): factor 1.5 / 1 / 3 alike, 0.5 and 0.25 lose rounds to missed spines.
__global__ void k_inject_scan_top(dfta::Job* __restrict__ jobs, const dfta::Job* __restrict__ scanned, const int* __restrict__ live, int nlive,
                                  const unsigned long long* __restrict__ scan_counters, double factor, double shift)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nlive || scan_counters[3] != 0ull) return;          // a sweep the scan could not decide / a lost member: no prediction
    const int k = live[q];
    if (jobs[k].phase != PH_TOP || jobs[k].frozen) return;
    const double T = scanned[k].top * (1.0 + shift);             // (shift: tests -- a recklessly wrong prediction changes rounds, never results)
    if (!(fabs(T) < 1e300)) return;
    // (scanned.bottom: the half width the scan itself left around T -- 0 from the group search, which bisects to the end)
    const double w = factor * (6e-11 * fabs(T) + 6e-10) + scanned[k].bottom - 1e-10 * fabs(T);
    jobs[k].hist_c[0] = T;
    jobs[k].hist_w[0] = w;
    jobs[k].hist_d[0] = 0.0;
    if (jobs[k].hist_ok < 2) jobs[k].hist_ok = 2;
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
    dfta_persist_destroy(&pb);
    persist_ok = false;
    if (d_jobs_scan) (void)hipFree(d_jobs_scan);
    if (d_counters_scan) (void)hipFree(d_counters_scan);
    d_jobs_scan = nullptr; d_counters_scan = nullptr;
    if (d_scan_live) (void)hipFree(d_scan_live);
    if (d_scan_xch) (void)hipFree(d_scan_xch);
    d_scan_live = nullptr; d_scan_xch = nullptr;
    if (d_jmatched) (void)hipFree(d_jmatched);
    if (d_jstart_keep) (void)hipFree(d_jstart_keep);
    d_jmatched = nullptr; d_jstart_keep = nullptr;
    own_ok = false;
    for (int** q : {&d_lane_job, &d_slot_off, &d_slot_jobs, &d_gsz, &d_goff, &d_pack_out, &d_live, &d_own_live, &d_queue}) { if (*q) (void)hipFree(*q); *q = nullptr; }
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
        if (const char* e = dfta_knob("LEVELS_NOISE")) sscanf(e, "%lf:%lf:%lf", &v[0], &v[1], &v[2]);   // "rel:abs:secant" of the noise band (':' -- the knob list itself is comma-separated)
        hist_extrapolate = dfta_knob("LEVELS_NOEXTRAP") == nullptr;
        if (const char* e = dfta_knob("LEVELS_EXTRAP")) sscanf(e, "%lf:%lf", &hist_kA, &hist_kB);
        if (const char* e = dfta_knob("LEVELS_SECANT_KAPPA")) k = atof(e);                                // trust in the parabolic correction
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_noise_rel), &v[0], sizeof(double));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_noise_abs), &v[1], sizeof(double));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_secant_noise), &v[2], sizeof(double));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_secant_kappa), &k, sizeof(double));
        const int fp = dfta_knob("LEVELS_NOFIXEDPOINT") ? 0 : 1;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fixed_point), &fp, sizeof(int));
        tuning[0] = v[0]; tuning[1] = v[1]; tuning[2] = v[2]; tuning[3] = k;
        fixed_point = fp;
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
        if (const char* e = dfta_knob("LEVELS_DEPTH")) d = atoi(e);                    // measurements: the automatic layout with another tree depth
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
    // the own-pace search of a batch (own.inc: one workgroup of W waves per live level, one launch): more than 64 live levels of an
    // un-chained solve on the logarithmic grid; the host rounds remain for everything else ($DFTA_DEBUG LEVELS_NOOWN, LEVELS_NOPERSIST: host rounds)
    own_ok = (tree_depth <= 0) && !dynamic && !g->uniform && mode == DFTA_LEVELS_BATCHED && ctx->sweep_kernel != DFTA_SWEEP_PIPELINED &&
             dfta_knob("LEVELS_OWN") != nullptr && dfta_knob("LEVELS_NOPERSIST") == nullptr && dfta_knob("LEVELS_STATIC") == nullptr;
    if (own_ok) {
        own_waves = 2048;                            // two waves per SIMD: what the sweep's registers (177 VGPRs) leave resident
        if (const char* e = dfta_knob("LEVELS_OWN_WAVES")) own_waves = std::max(64, atoi(e));      // measurements
        own_wmax = 8;
        if (const char* e = dfta_knob("LEVELS_OWN_WMAX")) own_wmax = std::min(8, std::max(1, atoi(e)));
        own_spine_cap = -1;
        if (const char* e = dfta_knob("LEVELS_OWN_SPINE_CAP")) own_spine_cap = atoi(e);
        ntrials = std::max<long>(ntrials, 64L * std::max<long>(own_waves, njobs));
    }
    nwaves = static_cast<int>(ntrials / 64);
    early_match = (dynamic || can_switch) && !g->uniform && dfta_knob("LEVELS_NOEARLYMATCH") == nullptr;
    // the device-side search serves the latency regime (one atom, or the last live atoms of a batch) on the logarithmic grid
    persist_ok = (dynamic || can_switch) && !g->uniform && ctx->sweep_kernel != DFTA_SWEEP_FUSED && dfta_knob("LEVELS_NOPERSIST") == nullptr;

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
    // the device-side search takes up to 256 live levels: one workgroup per level and compute unit, two for as many levels as there are
    // compute units left (all of them up to 128 levels), the rest in the pool; LEVELS_PERSIST_WIDE=n: up to 64 <= n <= 256 levels (64: one
    // atom's worth, round 5 -- larger batches then run host rounds)
    persist_cap = 256;
    if (const char* e = dfta_knob("LEVELS_PERSIST_WIDE")) persist_cap = std::min(256, std::max(64, atoi(e)));
    if (persist_ok) { const int prc = dfta_persist_create(ctx, g, std::min(njobs, persist_cap), &pb); if (prc) return prc; }
    if (can_switch) ALLOC(d_live, int, 64);
    if (!g->uniform && dfta_knob("LEVELS_NOQUEUE") == nullptr) ALLOC(d_queue, int, kSweepQueueClasses + 1 + (size_t)kSweepQueueClasses * nwaves);
    if (own_ok) ALLOC(d_own_live, int, njobs);
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
    // Round 6: the scan search (scan.hip: 40 - 60 us per sweep instead of 4 ms) runs the FIRST bisection of every live level ahead of the
    // device-side exact search and hands its end point to plan_round as the bracket of the exact search's first spine -- speculation only,
    // as the history bracket it replaces: the exact sweeps at the reference's midpoints take every decision.  One atom / a few atoms
    // (<= 64 jobs: the solver whose every run is the device-side search); $DFTA_DEBUG LEVELS_NOSCANPREDICT: off.
    // Batches (host rounds): the same with one workgroup per level (k_scan_levels, which stops at a quarter of the band); the scan tables of
    // all the batch's slots must be affordable (<= 2 GB).
    scan_predict = (sweep_mode != DFTA_SWEEPS_TOLERANCE && dfta_scan_supported(g) && dfta_knob("LEVELS_NOSCANPREDICT") == nullptr && use_prediction &&
                    mode == DFTA_LEVELS_BATCHED && tree_depth <= 0 &&
                    ((persist_ok && dynamic) || (!dynamic && (size_t)nslots * N * sizeof(double) <= ((size_t)2 << 30) && dfta_knob("LEVELS_NOSCANPREDICT_BATCH") == nullptr))) ? 1 : 0;
    scan_predict_factor = 1.5; scan_predict_shift = 0.0;
    if (const char* e = dfta_knob("LEVELS_SCAN_PREDICT_W")) scan_predict_factor = atof(e);
    if (const char* e = dfta_knob("LEVELS_SCAN_PREDICT_SHIFT")) scan_predict_shift = atof(e);
    if (scan_predict) { ALLOC(d_jobs_scan, Job, njobs); ALLOC(d_counters_scan, unsigned long long, 4); }
    if ((sweep_mode == DFTA_SWEEPS_TOLERANCE || scan_predict) && dfta_scan_supported(g)) {      // scan.hip: interleaved tables + per-lane {min, max}
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
    bool scan_wanted = false;          // a live level whose history bracket is still wide (or missing): the batch predictor runs
    // this run's layout: the solver's own, or -- a batch most of whose atoms have finished -- latency mode over the live jobs
    std::vector<int> live;
    if (frozen && h_last.size() == jobs.size())
        for (int k = 0; k < njobs; ++k) if (!frozen[k]) live.push_back(k);
    const bool sw = can_switch && !chained && frozen && h_last.size() == jobs.size() && !live.empty() && live.size() <= 64;
    const bool dyn = dynamic || sw;
    const bool pk = packed && !sw;
    const bool early = early_match && dyn;
    // the live levels of a latency-mode run search on the device, each at its own pace (persist.inc), when every one of them can have two
    // 64-trial blocks of its own; the host rounds below are the fallback
    std::vector<int> plive;
    if (sw) plive = live;
    else for (int k = 0; k < njobs; ++k) if (!(frozen && frozen[k] && h_last.size() == jobs.size())) plive.push_back(k);
    // (wide: 65 .. persist_cap live levels of a batch whose host rounds -- the fallback -- are the solver's own static / packed layout)
    const bool wide = !dyn && !own_ok && persist_cap > 64 && (int)plive.size() > 64;          // (LEVELS_OWN: that search was asked for)
    const bool use_persist = persist_ok && (dyn || wide) && !chained && sweep_mode != DFTA_SWEEPS_TOLERANCE && !plive.empty() && (int)plive.size() <= pb.nlive_cap &&
                             pb.nblocks / (int)plive.size() >= (persist_cap > 128 ? 1 : 2) && debug_rounds == 0;
    // more than 64 live levels: every level at its own pace all the same, one workgroup of W waves each in ONE ordinary launch (own.inc)
    const bool use_own = own_ok && !dyn && !chained && sweep_mode != DFTA_SWEEPS_TOLERANCE && !plive.empty() && debug_rounds == 0;
    int own_W = 1;
    if (use_own) { while (own_W < own_wmax && (long)plive.size() * own_W * 2 <= own_waves) own_W *= 2; }
    std::vector<int> plevel(njobs, -1);
    if (use_persist || use_own) for (size_t q = 0; q < plive.size(); ++q) plevel[plive[q]] = (int)q;
    if (sw) {
        DFTA_HIP(ctx, hipMemcpyAsync(d_live, live.data(), sizeof(int) * live.size(), hipMemcpyHostToDevice, st));
        tables_dirty = true;
    } else if (tables_dirty && !use_own) {           // back from latency mode / the own-pace search: the solver's own block tables
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
        if (use_persist) { j.tbase = plevel[k] * pb.tmax; j.tcap = 0; }      // its own region of the device-side search's trial arrays
        if (use_own) { j.tbase = plevel[k] * 64 * own_W; j.tcap = 64 * own_W; }
        j.rounds = 0;
        j.t_end_us = 0;
        j.deep = -1;
        j.cand_hist = (use_prediction && h_last.size() == jobs.size() && dfta_knob("LEVELS_PERSIST_NOBUDGET") == nullptr) ? h_last[k].cand_hist : 0;
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
                constexpr double kHistSafety = 2.0;
                const double d = h.hist_ok >= 1 ? T[ph] - h.hist_T[ph] : 0.0;
                j.hist_T[ph] = T[ph];
                j.hist_s[ph] = d;
                j.hist_d[ph] = h.hist_ok >= 1 ? fabs(d) : -1.0;
                const double q = (h.hist_ok >= 2 && h.hist_s[ph] != 0.0) ? d / h.hist_s[ph] : NAN;
                j.hist_q[ph] = q;
                j.hist_c[ph] = T[ph];
                j.hist_w[ph] = kHistSafety * fabs(d);
                // geometric convergence: extrapolate with the last ratio; the bracket is as wide as the ratio has been unsteady
                if (hist_extrapolate && h.hist_ok >= 3 && std::isfinite(q) && std::isfinite(h.hist_q[ph]) && fabs(q) <= 1.0 && fabs(h.hist_q[ph]) <= 1.0) {
                    const double u = (hist_kA + hist_kB * fabs(q - h.hist_q[ph])) * fabs(d);
                    if (u < j.hist_w[ph]) { j.hist_c[ph] = T[ph] + q * d; j.hist_w[ph] = u; }
                }
            }
            j.hist_ok = std::min(h.hist_ok + 1, 3);
        } else {
            j.hist_ok = 0;
        }
        {   // does this level still want the scan predictor of its first spine?  (scan.hip:k_scan_levels: the same rule, per level)
            const double T = j.hist_c[0], band = 6e-11 * fabs(T) + 6e-10;
            if (!(j.hist_ok >= 2 && j.hist_d[0] >= 0 && j.hist_w[0] + 1e-10 * fabs(T) + 64e-12 <= 4096. * band)) scan_wanted = true;
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
    persist_deep_reserve = 0;
    std::vector<int> second_share;          // more than nblocks / 2 live levels: the levels that get a second workgroup of their own (the others have one)
    if (use_persist && h_last.size() == jobs.size() && !pb.equal_shares) {
        // Feedback (speculation only): the levels whose search ended last in the previous step get first call on the pool (persist_plan,
        // deep = 2: as many as the pool can give a second share to), those within 15 % of the last one may match candidate eigenvalues
        // (deep = 1), the rest have time to spare (deep = 0)
        // (ranked by the maximum over the last three steps: a level that gets its deeper trees ends earlier, and a one-step memory would
        // take them away again in the next step)
        if (persist_tend.size() != 3 * jobs.size()) persist_tend.assign(3 * jobs.size(), 0);
        for (int k : plive) {
            int* h = &persist_tend[3 * (size_t)k];
            if (h_last[k].t_end_us > 0) { h[2] = h[1]; h[1] = h[0]; h[0] = h_last[k].t_end_us; }
        }
        auto score = [&](int k) { const int* h = &persist_tend[3 * (size_t)k]; return std::max(h[0], std::max(h[1], h[2])); };
        std::vector<int> order;
        for (int k : plive) if (score(k) > 0) order.push_back(k);
        if (order.size() == plive.size()) {
            std::sort(order.begin(), order.end(), [&](int x, int y) { return score(x) > score(y); });
            const int equal = pb.nblocks / (int)plive.size();
            if (equal == 1) second_share.assign(order.begin(), order.begin() + std::min<size_t>(order.size(), pb.nblocks - (int)plive.size()));
            int pool = pb.nblocks - equal * (int)plive.size();
            for (int k : plive) if (jobs[k].nodes == 0 && equal >= 8) pool += equal / 2;
            const int ndeep = std::min<int>((int)order.size(), pool / std::max(equal - 1, 1));
            persist_deep_reserve = ndeep * (equal - 1);
            const int tmax_us = score(order[0]);
            for (size_t q = 0; q < order.size(); ++q)
                jobs[order[q]].deep = (int)q < ndeep ? 2 : (score(order[q]) > 0.85 * tmax_us ? 1 : 0);
            if (pb.want_trace) {
                fprintf(stderr, "persist feedback: pool %d, reserve %d; search ends of the previous step [us]:", pool, persist_deep_reserve);
                for (size_t q = 0; q < order.size(); ++q) fprintf(stderr, " %d:%d%s", order[q], score(order[q]), jobs[order[q]].deep == 2 ? "*" : "");
                fprintf(stderr, "\n");
            }
        }
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
    bool persisted = false;
    int persist_rounds = 0;
    float ms_persist = 0;
    if (!scan && !chained && scan_predict && scan_tb.tabv != nullptr && !plive.empty() && debug_rounds == 0 && ((use_persist && !wide) || scan_wanted)) {
        // the scan search of the first bisection as a predictor of the exact search's first spines (speculation only): a group of workgroups per
        // level ahead of the device-side search, one workgroup per level ahead of the host rounds of a batch
        int K = 1;
        if (use_persist) for (int cand : {15, 7, 3}) if ((long)plive.size() * cand <= ctx->num_cu) { K = cand; break; }
        const bool grouped = use_persist && K > 1;             // (a wide device-side search: one workgroup per level, as for the host rounds)
        if (grouped || !use_persist || wide) {
            rc = dfta_launch_scan_build_tab(ctx, g, scan_tb, dV, d_slot_v, d_slot_l);
            if (rc) return rc;
            DFTA_HIP(ctx, hipMemcpyAsync(d_jobs_scan, d_jobs, sizeof(Job) * njobs, hipMemcpyDeviceToDevice, st));
            DFTA_HIP(ctx, hipMemcpyAsync(d_scan_live, plive.data(), sizeof(int) * plive.size(), hipMemcpyHostToDevice, st));
            DFTA_HIP(ctx, hipMemsetAsync(d_counters_scan, 0, sizeof(unsigned long long) * 4, st));
            const int prc = grouped
                ? dfta_launch_scan_levels_group(ctx, g, d_jobs_scan, d_scan_live, (int)plive.size(), K, scan_tb, fixed_point, d_counters_scan, d_scan_xch, -1, nullptr, nullptr)
                : dfta_launch_scan_levels(ctx, g, d_jobs_scan, d_chain_off_b, njobs, 0, scan_tb, fixed_point, d_counters_scan, -1, nullptr, nullptr);
            if (prc == DFTA_OK) {
                hipLaunchKernelGGL(k_inject_scan_top, dim3(((int)plive.size() + 63) / 64), dim3(64), 0, st, d_jobs, d_jobs_scan, d_scan_live, (int)plive.size(), d_counters_scan, scan_predict_factor, scan_predict_shift);
                DFTA_CHECK_LAUNCH(ctx);
            } else if (prc != DFTA_ERR_NOT_CONVERGED) return prc;
        }
    }
    if (!scan && use_persist) {
        dfta_range r_p("dfta: level search on the device (persistent kernel: sweeps, walk, match, normalisation)");
        DFTA_HIP(ctx, hipMemsetAsync(d_jstart_keep, 0xff, sizeof(int) * njobs, st));      // -1: frozen (the live levels write their cut-off index)
        if (stats) DFTA_HIP(ctx, hipEventRecord(ev[0], st));
        int aborted = 0;
        const bool want_trace = pb.want_trace;
        // A level without nodes has no second bisection (walk_job takes it without a sweep): a third of the search less, so half of its equal
        // share goes to the pool, from which the levels whose phases need one decision more than their trees hold take deeper ones
        std::vector<int> share(plive.size());
        const int equal = pb.nblocks / (int)plive.size();
        const bool float_shares = !pb.equal_shares && equal >= 8;
        for (size_t q = 0; q < plive.size(); ++q) share[q] = (float_shares && jobs[plive[q]].nodes == 0) ? equal - equal / 2 : equal;
        if (equal == 1 && !pb.equal_shares) {
            // one workgroup per level leaves nblocks - nlive over: second workgroups (scouts, a tree one level deeper) for the levels whose
            // search ended last in the previous steps; without that feedback, for the first ones of the list
            if (second_share.empty()) for (size_t q = 0; q < plive.size() && (int)q < pb.nblocks - (int)plive.size(); ++q) share[q] = 2;
            else for (int k : second_share) share[plevel[k]] = 2;
        }
        rc = dfta_launch_levels_persist(ctx, g, &pb, d_jobs, plive.data(), (int)plive.size(), d_tab, d_bounds, d_Psi, d_Q, d_jstart_keep, d_counters, stats != nullptr,
                                        use_prediction ? 0 : 1, integ_rule, tuning, fixed_point, &persist_rounds, &aborted, want_trace ? &persist_trace : nullptr, share.data(), persist_deep_reserve);
        if (rc) return rc;
        if (stats) { DFTA_HIP(ctx, hipEventRecord(ev[1], st)); DFTA_HIP(ctx, hipEventSynchronize(ev[1])); DFTA_HIP(ctx, hipEventElapsedTime(&ms_persist, ev[0], ev[1])); }
        ++persist_runs;
        if (want_trace && !aborted) {
            // one line per closed round, in the order of the closings: time since the first record [us], level, rounds taken, then what was planned
            const size_t n = persist_trace.size() / 4;
            std::vector<size_t> order(n);
            for (size_t q = 0; q < n; ++q) order[q] = q;
            std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return persist_trace[4 * x] < persist_trace[4 * y]; });
            const unsigned long long t0 = n ? persist_trace[4 * order[0]] : 0;
            fprintf(stderr, "persist trace: %zu records, kernel %.3f ms\n", n, ms_persist);
            for (size_t q : order) {
                const unsigned long long *w = &persist_trace[4 * q];
                const int ph = (int)(w[2] >> 56);
                if (ph == PH_DONE + 1) fprintf(stderr, "  %9.1f us  job %2d round %2d  search ended\n", (w[0] - t0) * 0.01, (int)(w[1] >> 32), (int)(w[1] & 0xffffffff));
                else if (ph == PH_DONE) fprintf(stderr, "  %9.1f us  job %2d round %2d  DONE (matched, start %llu, candidate %d)\n", (w[0] - t0) * 0.01, (int)(w[1] >> 32), (int)(w[1] & 0xffffffff), w[3] & 0xffffffffull, (int)(signed char)((w[3] >> 32) & 0xff));
                else fprintf(stderr, "  %9.1f us  job %2d round %2d  next: phase %d done %2d blocks %3d spine %2d capz %5d tcap %5d cand %d/%d\n", (w[0] - t0) * 0.01, (int)(w[1] >> 32),
                             (int)(w[1] & 0xffffffff), ph, (int)((w[2] >> 32) & 0xff), (int)((w[2] >> 16) & 0xffff), (int)(w[2] & 0xffff), (int)((w[3] >> 32) & 0xffff), (int)(w[3] & 0xffffffff),
                             (int)((w[3] >> 56) & 0xff), (int)(signed char)((w[3] >> 48) & 0xff));
            }
        }
        if (!aborted) persisted = true;
        else {
            // a worker was lost (or the grid cannot be co-resident): the whole solve again with host rounds, from the records as they were
            ++persist_fallbacks;
            {   // said once per process: the results are the same, the step is slower (dfta_step_stats::levels_fallbacks counts them)
                static bool told = false;
                if (!told) {
                    told = true;
                    fprintf(stderr, "dftatom_hip: the device-side level search did not finish (a workgroup of its cooperative launch was not scheduled -- "
                                    "another process on the device, masked compute units -- or timed out); this solve and any later one that fails the "
                                    "same way run with host-synchronised rounds instead: same results, slower steps\n");
                }
            }
            for (int k = 0; k < njobs; ++k)
                if (!jobs[k].frozen) { jobs[k].tbase = (pk || sw) ? 0 : k * tpj; jobs[k].tcap = (pk || sw) ? 0 : tpj; }
            DFTA_HIP(ctx, hipMemcpyAsync(d_jobs, jobs.data(), sizeof(Job) * njobs, hipMemcpyHostToDevice, st));
            DFTA_HIP(ctx, hipMemsetAsync(d_counters, 0, sizeof(unsigned long long) * 4, st));
            if (!chained && clamp_bottoms) {
                hipLaunchKernelGGL(k_clamp_bottoms, dim3((njobs + 63) / 64), dim3(64), 0, st, d_jobs, njobs, d_slot_min);
                DFTA_CHECK_LAUNCH(ctx);
            }
        }
    }
    bool owned = false;
    if (!scan && !persisted && use_own) {
        dfta_range r_o("dfta: level search on the device (own pace: one workgroup per level, sweeps + walk)");
        DFTA_HIP(ctx, hipMemcpyAsync(d_own_live, plive.data(), sizeof(int) * plive.size(), hipMemcpyHostToDevice, st));
        if (stats) DFTA_HIP(ctx, hipEventRecord(ev[0], st));
        rc = dfta_launch_levels_own(ctx, g, d_jobs, d_own_live, (int)plive.size(), own_W, d_tab, d_bounds, d_wave_slot, d_wave_first, d_wave_cnt, d_E, d_limit, d_start,
                                    d_us, d_us1, d_count, d_u0, d_phi, d_istop, d_trip, d_counters, stats != nullptr, use_prediction ? 0 : 1, own_spine_cap);
        if (rc) return rc;
        if (stats) DFTA_HIP(ctx, hipEventRecord(ev[1], st));
        tables_dirty = true;             // the kernel wrote its blocks' table slots
        owned = true;
        own_last_W = own_W;
    }
    if (!scan && !persisted && !owned) {
        rc = plan();
        if (rc) return rc;
    }
    // trials of the coming round: the whole static / latency-mode layout, or what k_pack has just laid out
    long round_trials = dyn ? budget_trials : static_trials;
    int pack_out[4] = {0, 0, 0, 0};
    if (pk && !scan && !persisted && !owned) {
        DFTA_HIP(ctx, hipMemcpyAsync(pack_out, d_pack_out, sizeof(pack_out), hipMemcpyDeviceToHost, st));
        DFTA_HIP(ctx, hipStreamSynchronize(st));
        round_trials = pack_out[0];
    }
    if (early && !scan && !persisted && !owned) {
        hipLaunchKernelGGL(k_job_slots, dim3((njobs + 63) / 64), dim3(64), 0, st, d_jobs, njobs, d_jslot, d_jl);
        DFTA_CHECK_LAUNCH(ctx);
        DFTA_HIP(ctx, hipMemsetAsync(d_jE, 0xff, sizeof(double) * njobs, st));      // NaN: "no energy yet" (k_take_ready)
    }
    int* d_ndone = reinterpret_cast<int*>(d_counters + 2);
    int rounds = scan ? 1 : (persisted ? persist_rounds : 0);
    float ms_sweep = persisted ? ms_persist : ms_scan;
    const int max_rounds = 4096;
    int done_seen = nfrozen;
    bool early_pending = false;
    while (!scan && !persisted && !owned && rounds < max_rounds) {
        dfta_range r_round("dfta: level-search round (expand, sweeps, scout, walk, plan)");
        if (round_trials <= 0 || round_trials > ntrials) { snprintf(ctx->err, sizeof(ctx->err), "level solver: packed round of %ld trials (room for %ld)", round_trials, ntrials); return DFTA_ERR_HIP; }
        const int round_waves = static_cast<int>(round_trials / 64);
        // the fused sweeps of a batch are launched as a queue of blocks, longest first (numerov.hip:k_sweep_queue)
        // (the pipelined kernel likewise once a round has more blocks than compute units: its second pass is then made of the short blocks)
        const bool queued = d_queue != nullptr && !g->uniform && (dfta_sweep_is_fused(ctx, round_waves) || round_waves > ctx->num_cu);
        if (queued) DFTA_HIP(ctx, hipMemsetAsync(d_queue, 0, sizeof(int) * (kSweepQueueClasses + 1), st));
        hipLaunchKernelGGL(k_expand, dim3((unsigned)((round_trials + 255) / 256)), dim3(256), 0, st, d_jobs, pk ? d_lane_job : d_wave_job, pk ? 0 : 6,
                           (int)round_trials, g->d_r, N, g->delta, g->far_arg_threshold, d_E, d_limit, d_start, d_us, d_us1, d_wave_kind, d_counters, g->uniform,
                           g->Rmax, g->h, queued ? d_queue : nullptr, queued ? d_queue + kSweepQueueClasses + 1 : nullptr, nwaves);
        DFTA_CHECK_LAUNCH(ctx);
        if (stats) DFTA_HIP(ctx, hipEventRecord(ev[0], st));
        rc = dfta_launch_sweep(ctx, g, DFTA_SWEEP_COUNT, d_wave_kind, round_waves, d_tab, d_wave_slot, d_wave_first, d_wave_cnt, d_E,
                               d_limit, d_start, d_us, d_us1, d_count, d_u0, stats ? d_trip : nullptr, stats ? d_counters + 1 : nullptr, g->uniform ? nullptr : d_bounds, d_phi,
                               d_istop, d_slot_l, queued ? d_queue : nullptr, nwaves);
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
    if (persisted) {
        // every live level was matched and normalised by the workgroup that closed its search
    } else if (scan && scan_match_mode) {
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
        unsigned long long cnt[3];
        DFTA_HIP(ctx, hipMemcpyAsync(cnt, d_counters, sizeof(cnt), hipMemcpyDeviceToHost, st));
        DFTA_HIP(ctx, hipStreamSynchronize(st));
        if (owned) {
            rounds = (int)(cnt[2] & 0xffffffffull);                 // the most rounds a level took
            DFTA_HIP(ctx, hipEventElapsedTime(&ms_sweep, ev[0], ev[1]));
        }
        stats->rounds = rounds;
        stats->sweeps_issued = static_cast<long>(cnt[0]) + 2L * (njobs - nfrozen);    // + inward/outward halves of the match solve
        stats->points_traversed = static_cast<long>(cnt[1]);
        stats->ms_sweep = ms_sweep;
        stats->layout = owned ? 6 : (persisted ? 5 : (scan ? 4 : (sw ? 3 : (dynamic ? 1 : (pk ? 2 : 0)))));
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
