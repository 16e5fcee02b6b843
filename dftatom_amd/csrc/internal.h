// internal.h -- launch helpers shared between the translation units of libdftatom_hip (not part of the ABI).
#pragma once
#include "common.h"

// numerov.hip
// doubles2 per table slot in a `bounds` buffer: the fast-division bounds of the slot + { min, max } of veff per block of points
int dfta_bounds_stride(const dfta_grid* g);
int dfta_launch_build_tab(dfta_ctx* ctx, const dfta_grid* g, double2* tab, const double* dV, const int* d_slot_v,
                          const int* d_slot_l, int nslots, double2* bounds /* nslots, may be null */);
// for_match / dL / dUz: uniform grid only -- the match solve re-derives its step from the truncated step count, so its
// second start value differs from a sweep's, and it needs GetBoundaryValueZero(h', l) per trial (Numerov.h:430,475)
// stream: nullptr = the context's stream (the early match solves of levels.hip run on a second one)
int dfta_launch_boundary(dfta_ctx* ctx, const dfta_grid* g, const double* dE, int ntrials, int* dStart, double* dUs,
                         double* dUs1, int for_match = 0, const int* dL = nullptr, double* dUz = nullptr, hipStream_t stream = nullptr);
// flag in SweepArgs::istop (numerov.hip): the sweep left CountNodes because the count exceeded the limit
constexpr int kStopOver = 0x40000000;

int dfta_launch_sweep(dfta_ctx* ctx, const dfta_grid* g, int kind, const int* blk_kind, int nblocks, const double2* tab,
                      const int* blk_slot, const int* blk_first, const int* blk_cnt, const double* dE, const int* dLimit,
                      const int* dStart, const double* dUs, const double* dUs1, int* dCount, double* dU0, int* dTrip,
                      unsigned long long* dTotalTrips, const double2* bounds /* per slot, may be null */,
                      double* dPhi = nullptr, int* dIstop = nullptr /* SweepArgs::phi / istop, may be null */,
                      const int* d_slot_l = nullptr /* uniform grid: l per table slot */,
                      const int* d_queue = nullptr /* fused kernel: the round's work queue (k_expand of levels.hip), lists of qcap blocks */, int qcap = 0);
// Balanced launch of the fused sweeps: the blocks of a round are entered into kSweepQueueClasses lists by expected length, and the launch's
// waves (two per SIMD, all resident) take them longest first through one ticket counter -- see k_sweep_queue
constexpr int kSweepQueueClasses = 16;
bool dfta_sweep_is_fused(const dfta_ctx* ctx, int nblocks);
int dfta_launch_match(dfta_ctx* ctx, const dfta_grid* g, int ntrials, const double2* tab, const int* d_trial_slot,
                      const double* dE, const int* dStart, const double* dUs, const double* dUs1, const int* dL,
                      double* dPsi, double* dQ, int* dMatch, const double2* bounds /* per slot (dfta_bounds_stride), may be null */,
                      const double* dUz = nullptr /* uniform grid: start value at the first node per trial */, hipStream_t stream = nullptr);

// persist.inc (compiled with numerov.hip): the exact level search of up to 64 levels on the device, every level at its own pace
namespace dfta { struct Job; }
struct dfta_persist_buffers {
    void* d_ctl = nullptr;         // control block, mailboxes, per-level workgroup lists, trace
    size_t ctl_bytes = 0;
    double *E = nullptr, *us = nullptr, *us1 = nullptr, *u0 = nullptr, *phi = nullptr;      // trial arrays: nlive_cap x tmax
    int *limit = nullptr, *start = nullptr, *count = nullptr, *istop = nullptr, *trip = nullptr;
    double *candP = nullptr, *candQ = nullptr;     // per workgroup: wavefunction + scratch of a speculative match (null: none)
    int* blk = nullptr;            // per workgroup: table slot, first trial, trial count of its running block
    int nblocks = 0, tmax = 0, nlive_cap = 0, trace_cap = 0;
    // $DFTA_DEBUG knobs, read when the buffers are made (as every other knob of a solver)
    int fault_block = -1;          // FAULT_PERSIST_WORKER: this workgroup drops its first block (tests)
    double timeout_ms = 0;         // LEVELS_PERSIST_TIMEOUT_MS (0: from the grid size)
    bool plain_launch = false;     // LEVELS_PERSIST_PLAIN_LAUNCH: no cooperative launch
    bool equal_shares = false;     // LEVELS_PERSIST_EQUAL: node-less levels keep their whole share
    bool want_trace = false;       // LEVELS_PERSIST_TRACE
    std::vector<unsigned char> h_stage;
};
int dfta_persist_create(dfta_ctx* ctx, const dfta_grid* g, int nlive_cap, dfta_persist_buffers* pb);
void dfta_persist_destroy(dfta_persist_buffers* pb);
int dfta_launch_levels_persist(dfta_ctx* ctx, const dfta_grid* g, dfta_persist_buffers* pb, dfta::Job* d_jobs, const int* live, int nlive,
                               const double2* d_tab, const double2* d_bounds, double* d_Psi, double* d_Q, int* d_jstart_keep,
                               unsigned long long* d_counters, bool stats, int nopredict, int integ_rule, const double* tuning, int fixed_point,
                               int* rounds, int* aborted, std::vector<unsigned long long>* trace_out, const int* share = nullptr /* host, nlive: workgroups per level */,
                               int deep_reserve = 0 /* workgroups the pool keeps for the levels marked Job::deep == 2 */);

// own.inc (compiled with numerov.hip): the exact level search of a batch on the device, one workgroup of W waves per live level, one
// ordinary launch; d_counters[0] += issued trials, [1] += traversed points (stats), [2] = max rounds of a level (as unsigned int)
int dfta_launch_levels_own(dfta_ctx* ctx, const dfta_grid* g, dfta::Job* d_jobs, const int* d_live, int nlive, int W, const double2* d_tab, const double2* d_bounds,
                           int* blk_slot, const int* blk_first, const int* blk_cnt, double* dE, int* dLimit, int* dStart, double* dUs, double* dUs1, int* dCount,
                           double* dU0, double* dPhi, int* dIstop, int* dTrip, unsigned long long* d_counters, bool stats, int nopredict, int spine_cap);

// scan.hip: the tolerance mode of the sweeps (transfer-matrix scan: one workgroup per trial)
struct dfta_scan_tables {
    double* tabv = nullptr;    // nslots * N: veff = V + c_l per slot, lane-interleaved (row i = t C + k at k 512 + t, row N-1 at N-1)
    double2* mm = nullptr;     // nslots * 512: {min, max} of veff over a lane's rows
    double* Atop = nullptr;    // 513: A = 2 Rp^2 delta^2 exp(2 i delta) of every lane's top row (and of row N-1)
    double* T = nullptr;       // C: exp(-2 delta (C-1-k))
    // tabv and T point kScanPadRows rows / 8 entries INTO their allocations: the row loops issue the loads of the next batches
    // unconditionally (rows below row 0 of slot 0, entries below T[0]: read, never used) -- no clamps, no branches in the loops
    double* tabv_alloc = nullptr;
    double* T_alloc = nullptr;
    int nslots = 0;
};
int dfta_scan_supported(const dfta_grid* g);
int dfta_scan_tables_create(dfta_ctx* ctx, const dfta_grid* g, int nslots, dfta_scan_tables* tb);
void dfta_scan_tables_destroy(dfta_scan_tables* tb);
int dfta_launch_scan_build_tab(dfta_ctx* ctx, const dfta_grid* g, const dfta_scan_tables& tb, const double* dV, const int* d_slot_v, const int* d_slot_l);
int dfta_launch_scan_sweeps(dfta_ctx* ctx, const dfta_grid* g, int kind, int ntrials, const dfta_scan_tables& tb, const int* d_trial_slot,
                            const double* dE, const int* dLimit, int* dCount, double* dU0, int* dStart, int* dTrip, int* dBad);

// reduce.hip: Integral::{Trapezoid,SimpsonOneThird,Simpson38,Boole,Romberg} (Integral.h:11-155) with the reference's
// sequential summation order, one wave per vector: out[k] = rule(delta, vals + k*stride) for k < nvec
int dfta_launch_integrate_ordered(dfta_ctx* ctx, int rule /* DFTA_INT_* */, double delta, const double* dVals, int n, int nvec,
                                  size_t stride, double* dOut);
// Simpson 3/8 (Integral.h:50-73) with its two sums taken in parallel (tolerance mode of the sweeps: the normalisation integral of
// scan_match is summed that way too); same weights, a different order of additions
int dfta_launch_integrate_simpson38_parallel(dfta_ctx* ctx, double delta, const double* dVals, int n, int nvec, size_t stride, double* dOut);
int dfta_integral_shape_ok(int rule, int sz);
