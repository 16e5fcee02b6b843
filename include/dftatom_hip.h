/*
 * dftatom_hip.h -- C ABI of the MI355X (gfx950) radial-DFT inner loop.
 *
 * Drop-in boundary for the numerical core of aromanro/DFTAtom.  The reference has no FFI; its seam is
 * the public C++ surface of DFT::Numerov / DFT::PoissonSolver / DFT::VWNExchCor / DFT::Integral /
 * DFT::DFTAtom.  Every entry point below names the reference interface it replaces
 * (file:line under /root/reference/DFTAtom).  dftatom_amd/compat/ holds C++ classes with the reference's
 * names and signatures that forward to this ABI.
 *
 * Conventions
 *   - plain C types only; every function returns an int status (DFTA_OK == 0) and never throws;
 *   - all floating point is IEEE fp64; kernels are built with -ffp-contract=off and use only + - * / sqrt on
 *     values, with exp() tables built on the host exactly as the reference evaluates them;
 *   - `_dev` functions take DEVICE pointers and are asynchronous on the context's stream;
 *     functions without the suffix take HOST pointers, copy, run the same kernels and synchronise;
 *   - one in-flight call per context (the reference is single threaded: DFTAtomFrame.cpp:176-198);
 *   - there is NO CPU fallback: without a usable HIP device every call fails with DFTA_ERR_NO_DEVICE.
 */
#ifndef DFTATOM_HIP_H
#define DFTATOM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFTA_OK              0
#define DFTA_ERR_INVALID     1   /* bad argument */
#define DFTA_ERR_NO_DEVICE   2   /* no HIP device / HIP runtime failure at start-up */
#define DFTA_ERR_HIP         3   /* HIP runtime error, see dfta_last_error() */
#define DFTA_ERR_NOMEM       4
#define DFTA_ERR_NOT_CONVERGED 5 /* soft: DFTAtom.cpp:538-539 `didNotConverge` */

typedef struct dfta_ctx     dfta_ctx;      /* device + stream + scratch                              */
typedef struct dfta_grid    dfta_grid;     /* device-resident tables of one logarithmic radial grid   */
typedef struct dfta_poisson dfta_poisson;  /* multigrid level storage for a batch of atoms            */
typedef struct dfta_scf     dfta_scf;      /* device-resident SCF state of a batch of atoms           */

/* ---- context -------------------------------------------------------------------------------------- */
/* hip_stream: a hipStream_t to launch on (e.g. torch.cuda.current_stream().cuda_stream), or NULL for a
 * stream owned by the context. */
int         dfta_ctx_create(int device, void* hip_stream, dfta_ctx** out);
void        dfta_ctx_destroy(dfta_ctx* ctx);
int         dfta_ctx_synchronize(dfta_ctx* ctx);
const char* dfta_last_error(const dfta_ctx* ctx);
const char* dfta_version(void);
/* HIP-event duration (ms, on the context's stream) of the dominant kernel of the last host-pointer call */
int         dfta_ctx_last_kernel_ms(dfta_ctx* ctx, float* ms);
/* Which of the two (bit-identical) Numerov sweep kernels dfta_numerov_sweeps / dfta_solve_levels / dfta_scf_step launch:
 * AUTO picks the pipelined kernel (one workgroup per block of 64 trials) while a launch has few blocks and the fused
 * one-wave-per-block kernel when the machine is full.  Initial value: AUTO, or $DFTA_SWEEP_KERNEL = fused | pipe. */
#define DFTA_SWEEP_AUTO      0
#define DFTA_SWEEP_FUSED     1
#define DFTA_SWEEP_PIPELINED 2
int         dfta_ctx_set_sweep_kernel(dfta_ctx* ctx, int which);
/* number of compute units / name of the device behind the context (reporting only) */
int         dfta_ctx_device_info(const dfta_ctx* ctx, int* num_cu, char* name, int name_cap);
/* Measurement aid (no counterpart in the reference): attainable HBM bandwidth of the device, GB/s, from a copy (c = a) and a
 * triad (a = b + s c) kernel over arrays of `doubles_per_array` fp64 each (>= 2^20, even; use >= 2^27 to defeat the 256 MB
 * Infinity Cache), best of `reps` timed repetitions after one warm-up.  bench.py quotes it next to the 8 TB/s spec figure. */
int         dfta_ctx_measure_hbm(dfta_ctx* ctx, size_t doubles_per_array, int reps, double* copy_gbs, double* triad_gbs);

/* ---- grid ------------------------------------------------------------------------------------------
 * Replaces NumerovFunctionNonUniformGrid's constructor and position/exp evaluations (Numerov.h:76-101,
 * 181-184), PoissonSolver::GetNumberOfNodes (PoissonSolver.h:127-135) and FillRNonuniformR
 * (PoissonSolver.cpp:212-223).  N = 2^mg_levels + 1.  Tables exp(i d), exp(2 i d), exp(i d / 2), r_i and
 * l(l+1)/(r_i r_i)*0.5 are evaluated once on the host with libm in the reference's operation order and
 * uploaded, so device kernels never call exp() on grid quantities. */
int    dfta_grid_create(dfta_ctx* ctx, int mg_levels, double delta, double Rmax, dfta_grid** out);
/* Uniform grid r_i = i h, h = Rmax / (N - 1): replaces NumerovFunctionRegularGrid (Numerov.h:16-70), the h of
 * DFTAtom.cpp:66-68 and PoissonSolver::FillR (PoissonSolver.cpp:200-210).  Every entry point below accepts either kind
 * of grid: sweeps then follow the IsUniform() branches of Numerov.h:274-291,353-370,408-425 (cut-off radius
 * 200/sqrt(2|E|), recurrence with h^2), the Poisson solver runs with deltaGrid = 0 (SolvePoissonUniform,
 * PoissonSolver.h:20-49) and the SCF is CalculateUniformLDA / LSDA (DFTAtom.cpp:60-210, 646-844). */
int    dfta_grid_create_uniform(dfta_ctx* ctx, int mg_levels, double Rmax, dfta_grid** out);
int    dfta_grid_is_uniform(const dfta_grid* g);
void   dfta_grid_destroy(dfta_grid* g);
int    dfta_grid_num_nodes(const dfta_grid* g);
double dfta_grid_rp(const dfta_grid* g);
int    dfta_grid_get_r(const dfta_grid* g, double* r_host);           /* N doubles */
int    dfta_num_nodes(int mg_levels);                                  /* PoissonSolver.h:127-135 */

/* ---- batched Numerov sweeps ------------------------------------------------------------------------
 * One trial = one call of
 *   DFTA_SWEEP_COUNT  Numerov<NonUniform>::SolveSchrodingerCountNodes     (Numerov.h:272-349)
 *   DFTA_SWEEP_ZERO   Numerov<NonUniform>::SolveSchrodingerSolutionInZero (Numerov.h:351-401)
 * for (potential index, l, E[, nodesLimit]).  All trials of a call are integrated concurrently, one lane
 * per trial, lanes of a wavefront marching the grid index together.
 *
 * boundary values (GetMaxRadiusIndex / GetBoundaryValueFar, Numerov.h:103-136):
 *   DFTA_BOUNDARY_DEVICE  cut-off index and the two start values are computed on the device (device exp());
 *   DFTA_BOUNDARY_HOST    they are computed on the host with libm exactly as the reference does and
 *                         uploaded -- sweeps are then bit-identical to the reference's.
 * The _dev variant always uses DFTA_BOUNDARY_DEVICE unless start/us/us1 are given (non-NULL). */
#define DFTA_SWEEP_COUNT 0
#define DFTA_SWEEP_ZERO  1
/* how a sweep is integrated: EXACT = the reference's rounding sequence (numerov.hip); TOLERANCE = transfer-matrix scan (scan.hip) */
#define DFTA_SWEEPS_EXACT     0
#define DFTA_SWEEPS_TOLERANCE 1
#define DFTA_BOUNDARY_DEVICE 0
#define DFTA_BOUNDARY_HOST   1

int dfta_numerov_sweeps(dfta_ctx* ctx, const dfta_grid* g, int kind, int boundary,
                        int nV, const double* V,            /* nV potentials, nV*N doubles (host)      */
                        int ntrials, const int* vidx,       /* potential index per trial (NULL -> 0)   */
                        const int* l, const double* E, const int* nodesLimit /* COUNT only */,
                        int* count_out,                     /* COUNT: node count per trial             */
                        double* u0_out,                     /* ZERO: extrapolated u(0); COUNT: optional */
                        int* start_out, int* trip_out);     /* optional diagnostics: cut-off index, loop trips */

/* device-pointer form; trials must be grouped: `ngroups` groups, group k covers trials
 * [group_off[k], group_off[k+1]) which share (group_vidx[k], group_l[k]).  All pointers are device
 * pointers except the small group_* arrays (host). */
int dfta_numerov_sweeps_dev(dfta_ctx* ctx, const dfta_grid* g, int kind,
                            int nV, const double* dV,
                            int ngroups, const int* group_off, const int* group_vidx, const int* group_l,
                            const double* dE, const int* dLimit,
                            const int* dStart, const double* dUs, const double* dUs1,   /* NULL -> device boundary */
                            int* dCount, double* dU0, int* dStartOut, int* dTrip);

/* TOLERANCE MODE of the same two sweeps (opt-in; logarithmic grids of 12 .. 20 multigrid levels): the recurrence of Numerov.h:309-321
 * is linear in w, w_{i-1} = (2 + f_i/(1 - f_i/12)) w_i - w_{i+1}, so ONE trial is integrated by the 512 lanes of a workgroup as a
 * transfer-matrix scan (segment products, log-depth combine, a second pass for the sign changes) in ~25 us at 131 073 points instead
 * of 4 ms as a dependent chain.  Same cut-off index, start values (device exp) and exit rules as above; a different order of roundings:
 * node counts equal the exact kernels' except inside the round-off band of a count transition (a few 1e-12 |E| wide), u(0) agrees to
 * ~1e-9 relative away from its zeros.  count_out holds CountNodes' decision value min(count, nodesLimit + 1); trip_out the loop trips up
 * to the classical-turning-point exit (not the early return at count > nodesLimit).  fallback_out[t] != 0 (optional): the scan met a
 * non-finite value or f >= 12 -- the exact kernels must decide that trial (never the case for potentials of an SCF). */
int dfta_numerov_sweeps_scan(dfta_ctx* ctx, const dfta_grid* g, int kind, int nV, const double* V, int ntrials, const int* vidx,
                             const int* l, const double* E, const int* nodesLimit, int* count_out, double* u0_out,
                             int* start_out, int* trip_out, int* fallback_out);

/* Numerov<NonUniform>::SolveSchrodingerMatchSolutionCompletely (Numerov.h:403-504) for a batch of
 * (vidx, l, E); Psi_out: ntrials*N doubles; matchPoint_out: ntrials. */
int dfta_numerov_match(dfta_ctx* ctx, const dfta_grid* g, int boundary, int nV, const double* V,
                       int ntrials, const int* vidx, const int* l, const double* E,
                       double* Psi_out, long* matchPoint_out);

/* ---- a potential resident on the device ------------------------------------------------------------------------------------------
 * The reference's Numerov keeps a REFERENCE to the caller's Potential and re-reads it on every call (Numerov.h:69,186), and its L3 makes
 * ~2100 calls on one Numerov object per SCF step.  dfta_numerov_sweeps / _match copy the potential and build the slot tables on every
 * call; a dfta_potential does both once (tables for l = 0..3), so that a call costs its sweep.  dfta_potential_update keeps the
 * "re-read on every call" semantics: a host memcmp against the resident copy, re-upload and rebuild only when the caller's values changed.
 * sweep_mode: DFTA_SWEEPS_EXACT (host boundary values: bit-identical to the reference) or DFTA_SWEEPS_TOLERANCE (scan sweeps,
 * ~0.1 ms per call at 131 073 points instead of 4 ms; a trial the scan cannot decide is repeated on the exact kernels).
 * dftatom_amd/compat/Numerov.h holds one per DFT::Numerov object. */
typedef struct dfta_potential dfta_potential;
int  dfta_potential_create(dfta_ctx* ctx, const dfta_grid* g, const double* V /* N, host */, dfta_potential** out);
int  dfta_potential_update(dfta_potential* p, const double* V /* N, host */);
void dfta_potential_destroy(dfta_potential* p);
int  dfta_potential_sweeps(dfta_potential* p, int kind, int sweep_mode, int ntrials, const int* l, const double* E, const int* nodesLimit,
                           int* count_out, double* u0_out, int* start_out, int* trip_out);
int  dfta_potential_match(dfta_potential* p, int ntrials, const int* l, const double* E, double* Psi_out, long* matchPoint_out);

/* ---- eigenvalue search for a batch of (atom,spin) potentials -----------------------------------------
 * Replaces DFTAtom::LoopOverLevels + LocateInterval + NormalizeNonUniform + the density update of
 * CalculateNonUniformDensity (DFTAtom.cpp:36-56, 328-343, 493-604).  All bisections run on the device as
 * speculative bisection trees of depth `tree_depth` that reproduce the reference's midpoint sequence.
 *   mode DFTA_LEVELS_CHAINED : BottomEnergy handed from level to level as DFTAtom.cpp:541 (E-3); levels of one
 *                              potential run one after the other (bit-for-bit the reference's bisection path)
 *   mode DFTA_LEVELS_BATCHED : all levels run concurrently and un-chained; level (v, l) starts from
 *                              max(bottom0[v], min_i Veff_l(i)) -- no eigenvalue lies below the minimum of the
 *                              effective potential, and below it the node-count predicate of l >= 1 misfires
 *                              (SURVEY C.12), which is why plain -Z^2-1 is not safe for f levels.  Documented
 *                              deviation from DFTAtom.cpp:541: the bisection paths differ, eigenvalues agree with
 *                              the chained path to ~1e-10 relative (tests).  bottom_hint, if given, is used as is. */
#define DFTA_LEVELS_CHAINED 0
#define DFTA_LEVELS_BATCHED 1
/* OR-ed into `mode` of dfta_solve_levels: the TOLERANCE MODE of the sweeps (see dfta_numerov_sweeps_scan).  The same three bisections
 * with the same midpoints, every trial integrated by the transfer-matrix scan: one workgroup runs LocateInterval and the u(0)
 * bisection of its level from start to end on the device -- no rounds, no speculation.  Eigenvalues agree with the exact path to the
 * width of the round-off band of the reference's own predicates (gate 2e-11 |E| + 1e-11 Ha; sweeps_reference counts stay
 * the reference's except for decisions inside that band).  Grids of 12 .. 20 multigrid levels; dfta_scf_options::sweep_mode for the SCF. */
#define DFTA_LEVELS_SCAN_SWEEPS 0x10

typedef struct dfta_level_result {
    double E;            /* eigenvalue (level.E, DFTAtom.cpp:534)                       */
    double top, bottom;  /* interval returned by LocateInterval                          */
    int    n_count;      /* reference-equivalent CountNodes sweeps on the bisection path */
    int    n_zero;       /* reference-equivalent SolutionInZero sweeps                   */
    int    converged;    /* !didNotConverge                                              */
    int    matchPoint;
    int    status;       /* DFTA_LEVEL_* bits: HOW the u(0) bisection ended (below)      */
} dfta_level_result;
/* The reference folds every way its u(0) bisection (DFTAtom.cpp:517-539) can fail to meet `Top - Bottom < 1e-12 && |u(0)| < 1e15` into
 * one flag, didNotConverge.  status tells them apart:
 *   CONVERGED       the stop test was met (converged != 0, no other bit);
 *   ITERATION_CAP   the 500-iteration cap ended the loop;
 *   FIXED_POINT     ... and the interval had collapsed long before: a step left (Top, Bottom) as they were, i.e. the same midpoint, sweep and
 *                   decision repeated to the cap -- |u(0)| never came below 1e15 (the case at 1 048 577 nodes from the seventh SCF step on;
 *                   those repeats are counted in n_zero but not integrated);
 *   U0_NONFINITE    u(0) of the last trial was NaN or infinite (overflow of the inward solution). */
#define DFTA_LEVEL_CONVERGED     1
#define DFTA_LEVEL_ITERATION_CAP 2
#define DFTA_LEVEL_FIXED_POINT   4
#define DFTA_LEVEL_U0_NONFINITE  8

int dfta_solve_levels(dfta_ctx* ctx, const dfta_grid* g, int mode, int tree_depth,
                      int nV, const double* V, const double* bottom0 /* nV */,
                      const double* bottom_hint /* BATCHED: per level bracket start, NULL -> bottom0[v] */,
                      int nlevels, const int* vidx, const int* n, const int* l, const int* occ,
                      dfta_level_result* results,           /* nlevels */
                      double* newDensity /* nV*N, accumulated occ*Psi^2 (i < N-1), may be NULL */,
                      double* Eelectronic /* nV, may be NULL */,
                      double* Psi_out /* nlevels*N normalised, may be NULL */,
                      long* issued_sweeps /* optional: sweeps actually launched */);

/* ---- multigrid Poisson -------------------------------------------------------------------------------
 * Replaces DFT::PoissonSolver (PoissonSolver.h:15-171, PoissonSolver.cpp).  `batch` independent atoms are
 * solved concurrently, one workgroup per atom. */
int  dfta_poisson_create(dfta_ctx* ctx, const dfta_grid* g, int batch, dfta_poisson** out);  /* PoissonSolver.cpp:8-27 */
/* Smoother mode.  EXACT (default): every Gauss-Seidel sweep (PoissonSolver.cpp:40-64) equals the reference's sequential sweep bit
 * for bit (lanes start 96 / 112 nodes early).  TOLERANCE (opt-in): 32-node warm-ups -- a lane's start value then
 * carries ~1e-9 of the change its start node undergoes in that sweep; the cycle converges to the same discrete solution and
 * round-off floor (gates: U within 2e-9 Z of the exact solve -- observed 4e-10 .. 9e-10 Z, what the reference itself moves by
 * under FMA contraction --, SCF energies 1e-9 relative, eigenvalues 1e-8 Ha + 2e-9 |E|), at about 75 % of the time.
 * dfta_poisson_create takes the mode from the knob list, DFTA_DEBUG="POISSON_MODE=tolerance" (or adaptive); default EXACT. */
#define DFTA_POISSON_EXACT     0
#define DFTA_POISSON_TOLERANCE 1
/* DFTA_POISSON_ADAPTIVE: the tolerance mode's kernels, and the V-cycles stop where the cycle has reached its round-off floor -- the norm of
 * the last level-0 sweep has not fallen below 0.7 x the previous cycle's twice in a row (and is below 1e-3 of the first cycle's) -- instead
 * of at the reference's cap of 100 (PoissonSolver.cpp:185-197: its test ||dPhi|| < 1e-14 lies below that floor, so the reference always
 * runs to the cap; the floor is reached after 6 .. 8 cycles).  Opt-in, never the default.  Gates against the exact 100-cycle
 * solve (tests/test_gpu_resident.py): U within 2e-9 Z on the logarithmic grids of the BASELINE configurations (the tolerance mode's gate),
 * 1e-8 Z on the uniform and nearly uniform grid, 2e-8 Z at 2^20+1 nodes (the conditioning of those solves: the reference's own solve moves
 * by 1.4e-8 relative under a 1e-12 perturbation there); the end residual is the exact solve's in every case. */
#define DFTA_POISSON_ADAPTIVE 2
int  dfta_poisson_create_ex(dfta_ctx* ctx, const dfta_grid* g, int batch, int mode, dfta_poisson** out);
int  dfta_poisson_mode(const dfta_poisson* p);
void dfta_poisson_destroy(dfta_poisson* p);
/* SolvePoissonNonUniform (PoissonSolver.h:51-81): density batch*N (host) -> U batch*N (host).
 * vcycles_out/err_out (optional, per atom): V-cycles executed (<=100) and last ||dPhi||_2. */
int  dfta_poisson_solve(dfta_poisson* p, const int* Z, const double* density, double* U,
                        int* vcycles_out, double* err_out);
/* device-pointer form; SYNCHRONISES: see dfta_poisson_group_info */
int  dfta_poisson_solve_dev(dfta_poisson* p, const int* dZ, const double* dDensity, double* dU);
/* While the batch leaves compute units idle, the fine levels of an atom are swept by a group of G workgroups that wait
 * for each other: the solve is launched cooperatively (the runtime refuses a grid that cannot be co-resident), the
 * barriers' spins are bounded, and after EVERY solve (dfta_poisson_solve, _solve_dev, dfta_scf_create, dfta_scf_step)
 * the groups' abort flag is inspected.  A refused launch or an aborted solve is repeated in the same process with one
 * workgroup per atom (bit-identical results), and the solver stays `degraded` to that path; `aborts` counts the solves
 * that had to be repeated.  $DFTA_FAULT_POISSON_MEMBER=1 (tests) makes the last member of every group return at once. */
int  dfta_poisson_group_info(const dfta_poisson* p, int* G, int* degraded, int* aborts);
/* unit-parity hooks on level storage of atom 0 (GaussSeidel / Restrict / Prolong / VCycle,
 * PoissonSolver.cpp:40-64, 110-157; PoissonSolver.h:155-159) */
int  dfta_poisson_level_size(const dfta_poisson* p, int lvl);
int  dfta_poisson_set_level(dfta_poisson* p, int lvl, const double* Phi, const double* Src);
int  dfta_poisson_get_level(dfta_poisson* p, int lvl, double* Phi, double* Src);
int  dfta_poisson_gauss_seidel(dfta_poisson* p, int lvl, int sweeps, double* err_out /* per sweep */);
/* IterateGaussSeidel (PoissonSolver.cpp:66-77): up to iterno sweeps, stops after a sweep with err < errorMin;
 * three sweeps of a chunked level run as one fused pass (same arithmetic, one third of the memory traffic) */
int  dfta_poisson_iterate_gs(dfta_poisson* p, int lvl, double errorMin, int iterno, double* err_out, int* sweeps_out);
int  dfta_poisson_restrict(dfta_poisson* p, int lvl);
int  dfta_poisson_prolong(dfta_poisson* p, int lvl_src);
int  dfta_poisson_vcycle(dfta_poisson* p, double* err_out);
/* PoissonSolver::SetBoundaries + FullCycle (PoissonSolver.cpp:29-33, PoissonSolver.h:89-124) on atom 0's level storage: the
 * level-0 source is the one the last solve (or dfta_poisson_set_level) left there */
int  dfta_poisson_full_cycle(dfta_poisson* p, double lowBoundary, double highBoundary, double errorMin, double errorMinLast,
                             double* err_out, int* vcycles_out);

/* ---- VWN exchange-correlation ---------------------------------------------------------------------------
 * VWNExchCor::Vexc / eexcDif, LDA (VWNExcCor.h:73-128) and LSDA (VWNExcCor.h:134-312). Host pointers. */
int dfta_vwn_lda(dfta_ctx* ctx, const double* n, size_t sz, double* vexc, double* eexcdif);
int dfta_vwn_lsda(dfta_ctx* ctx, const double* na, const double* nb, size_t sz,
                  double* vexc, double* va, double* vb, double* eexcdif);

/* Chachiyo's correlation functional with Dirac exchange, LDA: ChachiyoExchCor<Param>::Vexc / eexcDif (ExcCor.h:27-95);
 * improved != 0 selects ChachiyoExchCorImprovedParam (ExcCor.h:21-26).  The reference keeps it beside VWN with every
 * call site commented out (DFTAtom.cpp:383,412,421). */
int dfta_chachiyo_lda(dfta_ctx* ctx, int improved, const double* n, size_t sz, double* vexc, double* eexcdif);

/* ---- quadrature --------------------------------------------------------------------------------------------
 * Integral::{Trapezoid,SimpsonOneThird,Simpson38,Boole,Romberg} (Integral.h:11-155); values: host. */
#define DFTA_INT_TRAPEZOID 0
#define DFTA_INT_SIMPSON13 1
#define DFTA_INT_SIMPSON38 2
#define DFTA_INT_BOOLE     3
#define DFTA_INT_ROMBERG   4
int dfta_integrate(dfta_ctx* ctx, int rule, double delta, const double* values, int sz, double* result);

/* ---- SCF on a batch of atoms ---------------------------------------------------------------------------------
 * Replaces the body of DFTAtom::CalculateNonUniformLDA / LSDA (DFTAtom.cpp:346-491, 847-1022): state lives
 * in HBM between steps; one call advances every atom of the batch by one SCF iteration. */
typedef struct dfta_energies {
    double Etotal, Ekinetic, Ecoul, Enuclear, Exc;     /* as printed at DFTAtom.cpp:472 */
    double Eelectronic, Ehartree, eExcDif, Epotential;
} dfta_energies;

typedef struct dfta_step_stats {
    int    struct_size;          /* IN: sizeof(dfta_step_stats) of the CALLER's header (set it before every dfta_scf_step); the library
                                    fills no more than that many bytes, and rejects a value smaller than the version-6 prefix */
    int    levels_fallbacks;     /* solves of this step's level search that were repeated on another path: a sweep the scan could not decide or a
                                    lost worker of the device-side search (both: host rounds on the exact kernels, same results) */
    long   sweeps_issued;        /* Numerov sweeps launched (speculative trees)            */
    long   sweeps_reference;     /* sweeps on the reference's bisection path (count+zero+match) */
    long   points_traversed;     /* grid points traversed by issued sweeps                  */
    long   vcycles;              /* Poisson V-cycles executed over the batch                */
    int    rounds;               /* bisection rounds (= launches of the sweep kernel)       */
    float  ms_levels, ms_poisson, ms_tail;   /* HIP-event times of the three phases       */
    float  ms_sweep_kernels;     /* HIP-event time summed over the sweep-kernel launches only */
    float  ms_poisson_kernel;    /* HIP-event time of the persistent multigrid kernel        */
    long   sweeps_reference_executed; /* sweeps_reference minus the CountNodes calls of node-less levels' second
                                    bisection ("count < 0": decided without integrating, levels.hip) */
    long   points_reference;     /* grid points traversed by the executed sweeps on the reference's path */
    int    levels_layout;        /* trial layout of this step's level search: 0 one block of 2^depth trials per job, 1 latency mode
                                    (slots re-allotted every round, <= 64 jobs), 2 packed rounds (batches), 3 latency mode over the
                                    live jobs of a batch most of whose atoms have finished, 4 scan search (tolerance mode of the sweeps), 5 device-side exact
                                    search (one persistent kernel, every level at its own pace, up to 256 live levels: persist.inc), 6 own-pace search of a batch (one
                                    workgroup per level in one launch: own.inc, opt-in $DFTA_DEBUG LEVELS_OWN) -- never changes a result */
    int    poisson_groups;       /* workgroups per atom of the multigrid solve of this step (33 / 17: resident groups of up to 7 / 15 atoms); the live atoms of
                                    a batch are solved by a solver of their size class (128 / 64 / 32 / 16 / 15 / 7 atoms) once that is smaller
                                    than the batch's own */
} dfta_step_stats;

int  dfta_scf_create(dfta_ctx* ctx, const dfta_grid* g, int lsda, int natoms, const int* Z,
                     double alpha, int levels_mode, int tree_depth, dfta_scf** out);   /* DFTAtom.cpp:351-394 / 852-906 */
/* The reference's compile-time alternatives as run-time options (NULL = what the reference runs). */
#define DFTA_XC_VWN               0   /* VWNExchCor (live in the reference)                                  */
#define DFTA_XC_CHACHIYO          1   /* ChachiyoExchCor<ChachiyoExchCorParam>, LDA only (ExcCor.h)           */
#define DFTA_XC_CHACHIYO_IMPROVED 2   /* ChachiyoExchCor<ChachiyoExchCorImprovedParam> (DFTAtom.cpp:383)      */
typedef struct dfta_scf_options {
    int struct_size;  /* sizeof(dfta_scf_options) of the CALLER's header: members beyond it keep their defaults, a value that is no valid
                         size of this struct (0, or what an older header had in this place) is rejected with DFTA_ERR_INVALID          */
    int integrator;   /* DFTA_INT_*: quadrature of the energy integrals and of the normalisation (default SIMPSON38) */
    int functional;   /* DFTA_XC_*                                                                                */
    int aufbau;       /* DFTA_AUFBAU_*                                                                            */
    int poisson_mode; /* DFTA_POISSON_EXACT (0, default) / DFTA_POISSON_TOLERANCE / DFTA_POISSON_ADAPTIVE; -1: as dfta_poisson_create (DFTA_DEBUG POISSON_MODE=...) */
    int sweep_mode;   /* DFTA_SWEEPS_EXACT (0, default) / DFTA_SWEEPS_TOLERANCE (scan sweeps, see DFTA_LEVELS_SCAN_SWEEPS)                */
} dfta_scf_options;
/* The option and statistics structs start with struct_size (since version 6) and grow at the END between versions of this header:
 * zero-initialise them, set struct_size = sizeof(...) -- every other member's 0 is the reference's behaviour -- and the library reads /
 * writes no more than the caller's struct holds.  dfta_abi_version() returns the DFTA_ABI_VERSION the library was built with. */
#define DFTA_ABI_VERSION 6
int  dfta_abi_version(void);
int  dfta_scf_create_ex(dfta_ctx* ctx, const dfta_grid* g, int lsda, int natoms, const int* Z, double alpha, int levels_mode,
                        int tree_depth, const dfta_scf_options* options, dfta_scf** out);
void dfta_scf_destroy(dfta_scf* s);
int  dfta_scf_step(dfta_scf* s, dfta_step_stats* stats);                               /* DFTAtom.cpp:396-484 / 908-1009 */
/* An atom that has met the reference's stop test (DFTAtom.cpp:474-479: |dE/E| < 1e-11 and all levels converged in two
 * consecutive steps) is FROZEN: later steps of the batch leave its density, potential, eigenvalues, energies, step count
 * and record untouched, exactly as if it had been run alone, and cost nothing for it.
 * results of the last step (host copies): per atom energies; finished flag (the reference's Finished! test) */
int  dfta_scf_get_energies(dfta_scf* s, dfta_energies* e /* natoms */, int* finished /* natoms */);
/* geometry of the level search: bisection-tree depth, number of (atom,spin,n,l) jobs, trial lanes per round */
int  dfta_scf_info(const dfta_scf* s, int* tree_depth, int* njobs, long* trials_per_round);
int  dfta_scf_poisson_info(const dfta_scf* s, int* G, int* degraded, int* aborts);
/* Quadrature rule (DFTA_INT_*) of the live path: the five energy integrals of a step and the normalisation of every
 * orbital.  The reference calls Integral::Simpson38 at all 22 sites (DFTAtom.cpp:27,51,459-467,...) although its README
 * names Romberg (README.md:81); Simpson38 is the default, the other rules of Integral.h:11-155 are a switch away. */
int  dfta_scf_set_integrator(dfta_scf* s, int rule);   /* dfta_poisson_group_info of the SCF's solver */
int  dfta_scf_num_levels(const dfta_scf* s, int atom, int spin);
int  dfta_scf_get_levels(dfta_scf* s, int atom, int spin, int* n, int* l, int* occ, double* E, int* converged);
/* DFTA_LEVEL_* bits of every level of (atom, spin) after the last step, and the reference-equivalent sweep counts (any pointer may be NULL) */
int  dfta_scf_get_level_status(dfta_scf* s, int atom, int spin, int* status, int* n_count, int* n_zero);
int  dfta_scf_get_array(dfta_scf* s, int atom, int which, double* out /* N */);  /* 0 density,1 densityA,2 densityB,3 potA,4 potB,5 U */
/* fixed-size per-atom record for the periodic-table gather: 64 doubles */
#define DFTA_RECORD_DOUBLES 64
int  dfta_scf_get_records_dev(dfta_scf* s, double* dRecords /* natoms*64, device */);

/* ---- Aufbau -------------------------------------------------------------------------------------------------
 * AufbauPrinciple::GetSubshells + sort (AufbauPrinciple.h:36-75, DFTAtom.cpp:367); integer-only host code. */
int dfta_get_subshells(int Z, int* n, int* l, int* occ, int cap);
int dfta_split_spin(int Z, int* nA, int* nB, int* an, int* al, int* aocc, int* bn, int* bl, int* bocc, int cap); /* DFTAtom.cpp:611-638 */
/* aufbau: DFTA_AUFBAU_REFERENCE = what the reference runs (Madelung order + the f-block exceptions: Cr 3d4 4s2, Cu 3d9 4s2);
 * DFTA_AUFBAU_TRANSITION_METALS additionally applies AufbauPrinciple::AdjustForTransitionMetals (AufbauPrinciple.h:78-99),
 * which the reference defines but never calls (Cr 3d5 4s1, Cu 3d10 4s1, Pd 4d10, Pt 5d9 6s1 ...). */
#define DFTA_AUFBAU_REFERENCE          0
#define DFTA_AUFBAU_TRANSITION_METALS  1
int dfta_get_subshells_ex(int Z, int aufbau, int* n, int* l, int* occ, int cap);
int dfta_split_spin_ex(int Z, int aufbau, int* nA, int* nB, int* an, int* al, int* aocc, int* bn, int* bl, int* bocc, int cap);

#ifdef __cplusplus
}
#endif
#endif
