/*
 * dfta_oracle.h -- CPU ORACLE for the radial-DFT hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the arithmetic of aromanro/DFTAtom's numerical core
 * (Numerov shooting, level driver, multigrid Poisson, VWN, Newton-Cotes/Romberg quadrature,
 * Aufbau filling, one SCF step; since round 2 also the uniform-grid Numerov / level driver / Poisson).  Every function cites the reference file:line it follows.
 * It exists to CHECK the HIP path: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product library (dftatom_amd/csrc) never links, loads
 * or calls anything in this directory.
 *
 * Parity pin: oracle/ref_harness.cpp compiles the reference's own sources where they lie in
 * /root/reference into oracle/_ref/libdfta_ref.so; tests/test_oracle_vs_ref.py checks this
 * restatement against it bit-for-bit (sweeps, GS, restrict/prolong, VWN, quadrature, level
 * driver, full SCF steps) and tests/golden/ holds vectors generated from it.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 * -ffp-contract=off matters: the reference was pinned with g++ -O2 on x86-64 (no FMA).
 */
#ifndef DFTA_ORACLE_H
#define DFTA_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- grid (Numerov.h:76-87, DFTAtom.cpp:353-356) ------------------------------------- */
typedef struct dfo_grid {
    int    N;          /* number of nodes = 2^L + 1 */
    double delta;      /* deltaGrid */
    double Rmax;
    double Rp;         /* Rmax / (exp((N-1) delta) - 1) */
    double twodelta;   /* 2 delta */
    double Rp2delta2;  /* Rp^2 delta^2 */
    double delta2p4;   /* delta^2 / 4 */
} dfo_grid;

int    dfo_num_nodes(int levels);                                   /* PoissonSolver.h:127-135 */
void   dfo_grid_init(dfo_grid* g, int N, double delta, double Rmax); /* Numerov.h:76-87 */
double dfo_position(const dfo_grid* g, long i);                      /* Numerov.h:181-184 */
/* "table" variant of the CPU baseline: r_i and exp(2 i delta) looked up instead of re-evaluated (bit-identical results) */
void   dfo_tables_enable(const dfo_grid* g);
void   dfo_tables_disable(void);
double dfo_veff(const dfo_grid* g, const double* V, unsigned l, long i);        /* Numerov.h:89-94 */
double dfo_f(const dfo_grid* g, const double* V, unsigned l, double E, long i); /* Numerov.h:96-101 */
double dfo_far(const dfo_grid* g, double position, double E);        /* Numerov.h:103-108 */
double dfo_zero(const dfo_grid* g, double position, unsigned l);     /* Numerov.h:110-116 */
long   dfo_max_radius_index(const dfo_grid* g, double E, long maxIndex); /* Numerov.h:119-136 */

/* ---- Numerov sweeps (Numerov.h:272-504) ---------------------------------------------- */
/* returns node count; *trip (optional) = number of loop iterations executed, *start = cut-off index */
int    dfo_count_nodes(const dfo_grid* g, const double* V, unsigned l, double E, long nodesLimit,
                       long* start, long* trip);                     /* Numerov.h:272-349 */
double dfo_solution_in_zero(const dfo_grid* g, const double* V, unsigned l, double E,
                            long* start);                            /* Numerov.h:351-401 */
/* Precision yardstick of tests/test_scan_precision.py (no reference counterpart): u(0) of the inward sweep from cut-off index s in
 * three arithmetics -- variant 0: the reference's recurrence w_{i-1} = 2 w_i - w_{i+1} + u_i f_i in double (Numerov.h:309-321, 398),
 * 1: the same recurrence, its tables and start values in long double (80-bit on x86-64), 2: the summed form that
 * dftatom_amd/csrc/scan.hip integrates (D_{i-1} = D_i + g_i w_i, w_{i-1} = w_i + D_{i-1}, g = f / (1 - f/12)) in double. */
double dfo_u0_yardstick(const dfo_grid* g, const double* V, unsigned l, double E, long s, int variant);
/* Psi must hold N doubles; returns matchPoint */
long   dfo_match(const dfo_grid* g, const double* V, unsigned l, double E, double* Psi,
                 long* start);                                       /* Numerov.h:403-504 */

/* ---- level driver (DFTAtom.cpp:36-56, 328-343, 493-604) -------------------------------- */
typedef struct dfo_level {
    int    n;      /* m_N: 0-based principal index (1s -> 0) */
    int    l;      /* m_L */
    int    occ;    /* m_nrElectrons */
    double E;
    /* diagnostics filled by the driver */
    double top, bottom;       /* interval returned by LocateInterval */
    int    n_count, n_zero;   /* sweeps issued for this level */
    int    converged;
    long   matchPoint;
} dfo_level;

void dfo_locate_interval(const dfo_grid* g, const double* V, double* Top, double* Bottom,
                         int L, int NumNodes, double energyErr, int* ncalls); /* DFTAtom.cpp:566-604 */
void dfo_normalize_nonuniform(const dfo_grid* g, double* Psi);       /* DFTAtom.cpp:36-56 */
/* chained == 1: BottomEnergy = E-3 hand-over between levels exactly as DFTAtom.cpp:541 (the reference).
 * chained == 0: every level starts from the caller's BottomEnergy (study only: breaks f levels).
 * chained == 2: level k starts from hints[k] -- the batched GPU mode, hints[k] = E_{k-1}(previous SCF step) - 3. */
int  dfo_loop_over_levels(const dfo_grid* g, const double* V, dfo_level* levels, int nlevels,
                          double* newDensity, double* Eelectronic, double* BottomEnergy,
                          int chained, const double* hints);         /* DFTAtom.cpp:493-563 */
/* level-parallel variant of the un-chained clamped mode (bench.py's CPU baseline): see dfta_oracle.c */
void dfo_set_level_threads(int n);
int  dfo_get_level_threads(void);
int  dfo_calculate_density(const dfo_grid* g, const double* V, dfo_level* levels, int nlevels,
                           double* density, double alpha, double* newDensity,
                           double* Eelectronic, double BottomEnergy, int chained,
                           const double* hints);                     /* DFTAtom.cpp:328-343 */

/* ---- multigrid Poisson (PoissonSolver.h / PoissonSolver.cpp) ---------------------------- */
typedef struct dfo_poisson {
    int      levels;
    double   deltaGrid;
    int*     n;          /* n[l] nodes, l=0 finest */
    double** Phi;
    double** Src;
    double*  dlev;       /* deltaGridLevel */
    double   lowB, highB;
    /* counters (diagnostics) */
    long     n_gs, n_restrict, n_prolong, n_vcycles;
} dfo_poisson;

dfo_poisson* dfo_poisson_create(int levels, double dGrid);           /* PoissonSolver.cpp:8-27 */
void   dfo_poisson_destroy(dfo_poisson* p);
double dfo_gauss_seidel(dfo_poisson* p, int lvl);                    /* PoissonSolver.cpp:40-64 */
double dfo_iterate_gs(dfo_poisson* p, int lvl, double errorMin, int iterno); /* :66-77 */
void   dfo_restrict(dfo_poisson* p, int lvl);                        /* :126-157 */
void   dfo_prolong(const double* src, int nsrc, double* dst);        /* :110-123 */
void   dfo_initialize(dfo_poisson* p, double errorMin);              /* :80-106 */
double dfo_vcycle(dfo_poisson* p, double errorMin, int iterno);      /* PoissonSolver.h:155-159 */
double dfo_full_cycle(dfo_poisson* p, double errorMin, double errorMinLast); /* PoissonSolver.h:89-124 */
/* U must hold n[0] doubles */
double dfo_solve_poisson_nonuniform(dfo_poisson* p, int Z, double maxRadius,
                                    const double* density, double* U); /* PoissonSolver.h:51-81 */

/* ---- VWN (VWNExcCor.h, ExcCorBase.h) ---------------------------------------------------- */
void dfo_vwn_vexc(const double* n, double* out, size_t sz);          /* VWNExcCor.h:73-101 */
void dfo_vwn_eexcdif(const double* n, double* out, size_t sz);       /* VWNExcCor.h:103-128 */
void dfo_vwn_vexc_lsda(const double* na, const double* nb, double* res, double* va, double* vb,
                       size_t sz);                                   /* VWNExcCor.h:134-240 */
void dfo_vwn_eexcdif_lsda(const double* na, const double* nb, double* res, size_t sz); /* :242-312 */

/* ---- quadrature (Integral.h) ------------------------------------------------------------- */
double dfo_trapezoid(double delta, const double* v, int sz);         /* Integral.h:11-23 */
double dfo_simpson13(double delta, const double* v, int sz);         /* Integral.h:25-48 */
double dfo_simpson38(double delta, const double* v, int sz);         /* Integral.h:50-73 */
double dfo_boole(double delta, const double* v, int sz);             /* Integral.h:75-104 */
double dfo_romberg(double delta, const double* v, int sz, double err, int minSteps); /* :106-155 */

/* ---- Aufbau (AufbauPrinciple.h) ------------------------------------------------------------ */
/* fills levels (capacity >= 32) sorted by (N,L); returns count */
int dfo_get_subshells(int Z, dfo_level* levels);                     /* AufbauPrinciple.h:36-75 + sort DFTAtom.cpp:367 */
/* LSDA split (DFTAtom.cpp:611-638) */
void dfo_initialize_levels(int Z, int* nAlphaE, int* nBetaE, dfo_level* la, int* nla,
                           dfo_level* lb, int* nlb);

/* ---- SCF (DFTAtom.cpp:346-491 LDA, 847-1022 LSDA) ------------------------------------------ */
typedef struct dfo_energies {
    double Etotal, Ekinetic, Ecoul, Enuclear, Exc;   /* as printed at DFTAtom.cpp:472 */
    double Eelectronic, Ehartree, eExcDif, Epotential;
} dfo_energies;

typedef struct dfo_scf {
    int lsda;
    int Z, mgLevels;
    double alpha, MaxR, deltaGrid;
    dfo_grid g;
    dfo_poisson* ps;
    int nla, nlb;
    dfo_level la[32], lb[32];          /* LDA uses la only */
    double *density, *densityA, *densityB;
    double *potA, *potB;               /* LDA uses potA */
    double *U, *Vexc, *va, *vb, *eexc, *newDensity, *tmp[4];
    double Eold;
    int lastTimeConverged;
    int step;
    int finished;
    int chained;
} dfo_scf;

dfo_scf* dfo_scf_create(int lsda, int Z, int mgLevels, double alpha, double MaxR, double deltaGrid,
                        int chained);           /* set-up part: DFTAtom.cpp:351-394 / 852-906 */
void     dfo_scf_destroy(dfo_scf* s);
/* one iteration of the `for sp` loop body; returns 1 when the reference would print Finished! */
int      dfo_scf_step(dfo_scf* s, dfo_energies* e);   /* DFTAtom.cpp:396-484 / 908-1009 */

/* ---- uniform grid r_i = i h (Numerov.h:16-70 + IsUniform() branches, DFTAtom.cpp:21-33,213-325, PoissonSolver.h:20-49) ---- */
typedef struct dfo_ugrid { int N; double Rmax; double h; } dfo_ugrid;   /* h = Rmax / (N - 1) */
int    dfo_ucount_nodes(const dfo_ugrid* g, const double* V, unsigned l, double E, long nodesLimit, long* start);
double dfo_usolution_in_zero(const dfo_ugrid* g, const double* V, unsigned l, double E);
long   dfo_umatch(const dfo_ugrid* g, const double* V, unsigned l, double E, double* Psi);
void   dfo_normalize_uniform(double* Psi, int n, double h);
int    dfo_uloop_over_levels(const dfo_ugrid* g, const double* V, dfo_level* levels, int nlevels, double* newDensity,
                             double* Eelectronic, double* BottomEnergy);
double dfo_solve_poisson_uniform(dfo_poisson* p, int Z, double maxRadius, const double* density, double* U);

#ifdef __cplusplus
}
#endif
#endif
