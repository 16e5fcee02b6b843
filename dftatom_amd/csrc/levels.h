// levels.h -- device-resident eigenvalue search shared by dfta_solve_levels and the SCF driver.
#pragma once
#include <vector>

#include "common.h"
#include "internal.h"

namespace dfta {

struct JobSpec { int v, n, l, occ; };

// state of one (potential, level) eigenvalue search; lives in device memory
struct Job {
    double toe, boe;        // current interval of the running bisection (Top / Bottom)
    double bottom0;         // BottomEnergy at entry of LocateInterval
    double top, bottom;     // interval returned by LocateInterval
    double E;               // eigenvalue
    int v, n, l, occ, nodes, slot;
    int phase;
    int haveSgn, sgnBottom, iter3, converged;
    int n_count, n_zero, matchPoint;
    // path prediction (speculation only, never changes a decision): the decision strings of the three bisections in
    // the previous SCF step and how many of their leading bits held in the step before -- the next round evaluates
    // that many nodes along the predicted path (a "spine") and hangs the full speculative tree at its end
    unsigned long long pred_bits[3], cur_bits[3];
    int pred_len[3], cur_len[3], trust[3];
    int phase_done;         // decisions already taken in the running phase
    int spine;              // spine length of the trial layout of the current round
    // Predictions from the structure of the problem (speculation only).  A node enters the inward solution through
    // r = 0 exactly when u(0) changes sign, i.e. at an eigenvalue.  Hence (i) the energy at which the count reaches
    // `nodes` -- the end point of the SECOND bisection -- is where the count of the level with one node less (same
    // potential and l: job `sib`) exceeds nodes-1, which is the end point of ITS first bisection; (ii) the sign change
    // of the THIRD bisection lies at the upper end of its interval, the end point of the job's own first bisection.
    int sib;                // job of (v, n-1, l), or -1
    unsigned long long sp_bits;      // predicted decisions of the running phase (bit k = decision k of the phase) ...
    int sp_len;                      // ... valid for decisions < sp_len
    int use_sp;                      // the spine of this round follows sp_bits instead of pred_bits
    int miss;                        // a spine of the running phase mispredicted: no more spines in this phase
    // l > 0: the sign change of u(0) lies INSIDE the band between the two count thresholds (the count stops at the inner
    // turning point).  While the first two bisections run, the upper half of the job's trials (indices >= capz, whole
    // 64-trial blocks of kind ZERO) samples u(0) inside the part of the band that is already certain, then inside the
    // bracket of the sign change; the third bisection is predicted from that bracket.
    int capz;                        // trials [0, capz) = probe + spine + tree, [capz, tpj) = scouts (0: all tpj)
    double se_lo, se_hi;
    int se_state, se_sl, se_stop;    // bracketed? / sign of u(0) at se_lo / no longer shrinking
    double se_plo, se_phi;           // u(0) at se_lo / se_hi (NaN: unknown)
    double se_tlo, se_thi;           // secant estimate of the sign change inside the bracket, with its error bound
    int se_tok;
    // First bisection: every counted trial also reports where it stopped and how far the nearest zero of u was from that
    // point (SweepArgs::phi / istop, numerov.hip) -- a smooth function of E that crosses 0 where the count changes.  The
    // last sample on either side of the running bisection and the most recently replaced one give a secant estimate of
    // the end point with an error bound from the second divided difference: the next round's spine follows it.
    double sc_e[3], sc_phi[3];       // [0]: count <= nodes side (boe), [1]: count > nodes side (toe), [2]: replaced sample
    int sc_is[3];                    // istop of the samples (-1: none)
    double sc_lo, sc_hi;             // predicted bracket of the end point
    int sc_ok;
    // History: the end points of the three bisections in the previous SCF step (top, bottom, E) and how far each had moved
    // against the step before.  The potential changes by about half as much every step (mixing), so this step's end
    // point lies within a few times that distance of the last one: the spine follows the bisection while its midpoints
    // stay outside that bracket.  (Replaces the bit-prefix rule -- "as many leading decisions as matched last time" --
    // once two steps of history exist: a level close to a coarse bisection boundary flips an early decision now and then,
    // and a miss on the spine forfeits the round's tree.)
    double hist_T[3], hist_d[3];
    // Round 5: with linear mixing the end points converge geometrically -- the movement of one step is a nearly constant multiple q of the
    // previous one (Rn: q = 0.47 .. 0.49 from step 10 on, the extrapolated point T + q d is off by < 0.01 |d|; Cu LSDA alternates, q = -0.7).
    // hist_s: signed last movement, hist_q: its ratio to the one before (NaN: unknown); hist_c +- hist_w: the bracket the spine is
    // planned from -- T +- 2 |d| without ratios, T + q d +- (0.2 + 4 |q - q_prev|) |d| with them (levels.hip; 2 - 3 more predicted
    // decisions per first round than the two-step rule, profiles/history_brackets.py).
    double hist_s[3], hist_q[3], hist_c[3], hist_w[3];
    int hist_ok;                     // 0: none, 1: end points only, 2: end points and distances (3: and ratios)
    int frozen;                      // the job's atom has finished its SCF: the result of its last solve stands, nothing is integrated
    // Trial slots of the job in the current round: [tbase, tbase + tcap).  Static (k * tpj, tpj) when the launch fills the
    // machine; re-allotted every round by k_allot when a handful of jobs leave compute units idle (levels.hip).
    int tbase, tcap;
    int n_fixed;                     // iterations of the third bisection counted in n_zero but not integrated: the bisection stood on a fixed point (walk_job)
    int status;                      // DFTA_LEVEL_* bits: how the third bisection ended (DFTAtom.cpp:517-539)
    // device-side search (persist.inc): the certain part of the band (count == nodes) as the planner of the round saw it -- the scouts'
    // grid; the workers of a round must not read a sibling's record while it moves -- and the rounds this job has taken
    double sb_lo, sb_hi;
    int rounds;
    // feedback from one SCF step to the next (device-side search): when the level's search ended, in microseconds after the kernel's first
    // workgroup started; the host gives the levels that ended last in the previous step first call on the pool's workgroups (deep = 1)
    int t_end_us, deep;
    int cand_hist;                   // hit-rate budget of the level's candidate wavefunctions (persist.inc: cand_allowed), carried over by the host
    long long n_points;              // grid points traversed by the sweeps ON the bisection path (n_count + n_zero executed ones) + the match solve
};

struct LevelStats {
    int rounds = 0;
    long sweeps_issued = 0;
    long points_traversed = 0;
    float ms_sweep = 0;          // HIP-event time summed over the sweep launches
    int layout = 0;              // 0 static blocks, 1 latency mode, 2 packed rounds, 3 latency mode over the live jobs of a batch
};

struct LevelSolver {
    dfta_ctx* ctx = nullptr;
    const dfta_grid* g = nullptr;
    int mode = 0, nV = 0, njobs = 0, nchains = 0, nchains_chained = 0, depth = 0, tpj = 0, nwaves = 0, nslots = 0;
    long ntrials = 0;
    bool clamp_bottoms = true;     // BATCHED runs: bracket bottoms clamped to max(bottom, min Veff_l)
    std::vector<Job> h_jobs_template;
    std::vector<Job> h_last;       // job records of the previous solve (source of the path predictions)
    bool use_prediction = true;
    int debug_rounds = 0;          // $DFTA_DEBUG_ROUNDS, read once in setup()
    int integ_rule = DFTA_INT_SIMPSON38;   // quadrature of the normalisation integral (the reference calls Simpson38: DFTAtom.cpp:27,51)
    Job* d_jobs = nullptr;
    int *d_chain_off = nullptr, *d_chain_off_b = nullptr, *d_v_off = nullptr, *d_slot_v = nullptr, *d_slot_l = nullptr;
    double2* d_tab = nullptr;
    double *d_E = nullptr, *d_us = nullptr, *d_us1 = nullptr, *d_u0 = nullptr, *d_phi = nullptr;
    int *d_limit = nullptr, *d_start = nullptr, *d_count = nullptr, *d_istop = nullptr, *d_trip = nullptr;
    int *d_wave_kind = nullptr, *d_wave_slot = nullptr, *d_wave_first = nullptr, *d_wave_cnt = nullptr, *d_wave_job = nullptr;
    bool dynamic = false;          // trial slots re-allotted among the active jobs every round (few jobs: latency mode)
    // a static / packed solver whose live jobs have dropped to <= 64 (frozen atoms) runs its rounds in latency mode (k_allot over d_live)
    bool can_switch = false, tables_dirty = false;
    long static_trials = 0, budget_trials = 0;
    int* d_live = nullptr;
    std::vector<int> h_wave_job, h_wave_slot;
    // packed rounds (batches): the trials of a round laid out job after job inside their (table slot, kind) group by k_pack
    bool packed = false;
    int pack_dmin = 3, pack_dmax = 12, pack_lanes_small = 0, pack_dsmall = 3, pack_lanes_large = 0;
    int *d_lane_job = nullptr, *d_slot_off = nullptr, *d_slot_jobs = nullptr, *d_gsz = nullptr, *d_goff = nullptr, *d_pack_out = nullptr;
    unsigned long long* d_counters = nullptr;   // [0] issued trials, [1] traversed points, [2] scratch
    double *d_Psi = nullptr, *d_Q = nullptr;     // njobs*N each
    double *d_jE = nullptr, *d_jus = nullptr, *d_jus1 = nullptr;
    int *d_jslot = nullptr, *d_jl = nullptr, *d_jstart = nullptr, *d_jmp = nullptr;
    double* d_slot_min = nullptr;   // per table slot: min_i Veff_l(i)
    double2* d_bounds = nullptr;    // per table slot: fast-division range bounds (numerov.hip)
    hipEvent_t ev[2] = {nullptr, nullptr};
    // Early match solves (latency mode): a level whose search has ended is matched on a second stream while the remaining levels'
    // last rounds run -- the outer levels, whose outward streams are the long ones, end a round or two before the core levels.
    hipStream_t st2 = nullptr;
    hipEvent_t ev_walk = nullptr, ev_early = nullptr, ev_taken = nullptr;
    int *d_jmatched = nullptr, *d_jstart_keep = nullptr;
    double* d_snapE = nullptr;          // snapshot of the jobs' eigenvalues / "search ended" flags, taken on the first stream after every walk
    int *d_snapReady = nullptr, *d_jtake = nullptr;
    bool early_match = false;
    // Tolerance mode of the sweeps (scan.hip; DFTA_SWEEPS_TOLERANCE, set before setup()): interleaved tables per slot, per-lane {min, max}
    int sweep_mode = DFTA_SWEEPS_EXACT;
    dfta_scan_tables scan_tb;
    int* d_scan_live = nullptr;                   // live jobs of a grouped scan search
    unsigned long long* d_scan_xch = nullptr;     // 32 words per job: the members' results of a round, two parities
    int scan_group = 1;             // workgroups per level of the last scan search (1, 3, 7 or 15)
    int scan_predict = 0;           // $DFTA_DEBUG LEVELS_SCAN_PREDICT (measurements): the scan's first bisection predicts the exact search's first spines
    double scan_predict_factor = 1.5, scan_predict_shift = 0.0;
    Job* d_jobs_scan = nullptr;     // ... on a copy of the records
    unsigned long long* d_counters_scan = nullptr;
    int scan_fallbacks = 0;         // solves the scan handed back to the exact kernels (a trial it could not decide)
    // Device-side exact search (persist.inc): up to 64 live levels of an un-chained solve on the logarithmic grid run their three bisections in ONE
    // persistent kernel, every level at its own pace; the host rounds of run() remain for everything else and as the fallback
    bool persist_ok = false;        // the solver's shape allows it (decided in setup(); $DFTA_DEBUG LEVELS_NOPERSIST switches it off)
    dfta_persist_buffers pb;
    int persist_fallbacks = 0;      // solves repeated with host rounds after a lost worker
    int persist_deep_reserve = 0;   // workgroups of the pool kept for the levels that ended last in the previous steps
    std::vector<int> persist_tend;  // per job: 3 x the search-end times of the last three device-side solves [us] (0: none) -- the feedback ranks by their maximum
    int persist_runs = 0;
    // Own-pace search of a batch (own.inc): more than 64 live levels, one workgroup of W waves per level, one ordinary launch
    bool own_ok = false;
    int own_waves = 2048, own_wmax = 8, own_spine_cap = -1, own_last_W = 0;
    int* d_own_live = nullptr;
    // balanced launch of the fused sweeps (numerov.hip:k_sweep_queue): [0, C) entries per length class, [C] the ticket counter, then C lists of nwaves blocks
    int* d_queue = nullptr;
    double tuning[4] = {1e-11, 16e-12, 1.5e-11, 0.25};     // noise band (rel, abs, secant) and the secant's kappa, as set in setup()
    int fixed_point = 1;
    // history bracket of the first spines (Job::hist_c / hist_w): extrapolation with the last movement ratio ($DFTA_DEBUG LEVELS_NOEXTRAP: the
    // two-step rule only; LEVELS_EXTRAP="a:b": half width (a + b |q - q_prev|) |d|)
    bool hist_extrapolate = true;
    double hist_kA = 0.2, hist_kB = 4.0;
    std::vector<unsigned long long> persist_trace;          // $DFTA_DEBUG LEVELS_PERSIST_TRACE: 4 words per closed round of the last run

    LevelSolver() = default;
    LevelSolver(const LevelSolver&) = delete;
    LevelSolver& operator=(const LevelSolver&) = delete;
    ~LevelSolver();
    void release();
    int persist_cap = 256;      // live levels the device-side search takes (LEVELS_PERSIST_WIDE=n: 64 .. 256)
    int setup(dfta_ctx* c, const dfta_grid* grid, int mode, int tree_depth, int nV, const std::vector<JobSpec>& specs);
    // frozen (host, njobs, may be null): jobs whose result of the previous run() stands (finished atoms of an SCF batch)
    int run(const double* dV, const double* job_bottom, int run_mode, double* dNewDensity, LevelStats* stats,
            const unsigned char* frozen = nullptr);
    int fetch_jobs(std::vector<Job>& out);
};

}  // namespace dfta

// scan.hip: LocateInterval + the u(0) bisection of every chain of jobs by one workgroup each; counters[0] += executed sweeps,
// counters[1] += traversed points, counters[3] |= 1 when a sweep could not be decided by the scan
int dfta_launch_scan_levels(dfta_ctx* ctx, const dfta_grid* g, dfta::Job* d_jobs, const int* d_chain_off, int nchains, int chained,
                            const dfta_scan_tables& tb, int fixed_point, unsigned long long* d_counters,
                            int match_mode /* 0 none, 1 match, 2 match + Simpson-3/8 normalisation */, double* d_Psi, int* d_jstart_keep);
// K = 3, 7 or 15 workgroups per live job (un-chained brackets); DFTA_ERR_NOT_CONVERGED: the grid cannot be co-resident (use the one above)
int dfta_launch_scan_levels_group(dfta_ctx* ctx, const dfta_grid* g, dfta::Job* d_jobs, const int* d_live, int nlive, int K, const dfta_scan_tables& tb,
                                  int fixed_point, unsigned long long* d_counters, unsigned long long* d_xch, int match_mode, double* d_Psi, int* d_jstart_keep);
