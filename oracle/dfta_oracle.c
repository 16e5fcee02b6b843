/*
 * dfta_oracle.c -- CPU ORACLE (test infrastructure only; see dfta_oracle.h).
 *
 * Restates, in plain C and in the reference's operation order, the arithmetic of
 *   /root/reference/DFTAtom/{Numerov.h, DFTAtom.cpp, PoissonSolver.{h,cpp}, VWNExcCor.h,
 *                            ExcCorBase.h, Integral.h, AufbauPrinciple.h}.
 * Nothing here is used by the product path.  Pinned bit-for-bit against the compiled
 * reference (oracle/_ref) by tests/test_oracle_vs_ref.py and against tests/golden/.
 */
#define _GNU_SOURCE
#include "dfta_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static const double fourM_PI = 4. * M_PI;   /* PoissonSolver.h:12, DFTAtom.h:20 */

/* ======================================================================================= */
/* grid                                                                                    */
/* ======================================================================================= */

int dfo_num_nodes(int levels)   /* PoissonSolver.h:127-135, Ncoarse = 3 */
{
    int size = 3;
    for (int i = 0; i < levels - 1; ++i) size = size * 2 - 1;
    return size;
}

void dfo_grid_init(dfo_grid* g, int N, double delta, double Rmax)   /* Numerov.h:76-87 */
{
    g->N = N;
    g->delta = delta;
    g->Rmax = Rmax;
    g->Rp = Rmax / (exp(((double)N - 1.) * delta) - 1.);
    const double Rp2 = g->Rp * g->Rp;
    g->twodelta = 2. * delta;
    const double delta2 = delta * delta;
    g->Rp2delta2 = Rp2 * delta2;
    g->delta2p4 = delta2 * 0.25;
}

/* Optional "table" variant of the CPU baseline: the two exp() values per grid point that the
 * reference re-evaluates in every sweep are computed once per grid and looked up.  Same values, same arithmetic after
 * the lookup, hence bit-identical results (tests/test_oracle_golden.py); only the timing differs.  One grid at a time,
 * not thread safe (each cpu_baseline worker is its own process). */
static struct { int N; double delta, Rmax; double* pos; double* e2; } dfo_tab = {0, 0, 0, 0, 0};
void dfo_tables_enable(const dfo_grid* g)
{
    dfo_tables_disable();
    dfo_tab.pos = (double*)malloc(sizeof(double) * (size_t)g->N);
    dfo_tab.e2 = (double*)malloc(sizeof(double) * (size_t)g->N);
    for (long i = 0; i < g->N; ++i) {
        dfo_tab.pos[i] = g->Rp * (exp((double)i * g->delta) - 1.);
        dfo_tab.e2[i] = exp((double)i * g->twodelta);
    }
    dfo_tab.N = g->N; dfo_tab.delta = g->delta; dfo_tab.Rmax = g->Rmax;
}
void dfo_tables_disable(void)
{
    free(dfo_tab.pos); free(dfo_tab.e2);
    dfo_tab.pos = dfo_tab.e2 = 0; dfo_tab.N = 0;
}
static inline int dfo_tab_hit(const dfo_grid* g, long i)
{
    return dfo_tab.N == g->N && dfo_tab.delta == g->delta && dfo_tab.Rmax == g->Rmax && i >= 0 && i < g->N;
}

double dfo_position(const dfo_grid* g, long i)   /* Numerov.h:181-184 */
{
    if (dfo_tab_hit(g, i)) return dfo_tab.pos[i];
    return g->Rp * (exp((double)i * g->delta) - 1.);
}

double dfo_veff(const dfo_grid* g, const double* V, unsigned l, long i)   /* Numerov.h:89-94 */
{
    const double position = dfo_position(g, i);
    return V[i] + l * (l + 1.) / (position * position) * 0.5;
}

double dfo_f(const dfo_grid* g, const double* V, unsigned l, double E, long i)   /* Numerov.h:96-101 */
{
    const double effectivePotential = dfo_veff(g, V, l, i);
    if (dfo_tab_hit(g, i)) return 2. * (effectivePotential - E) * g->Rp2delta2 * dfo_tab.e2[i] + g->delta2p4;
    return 2. * (effectivePotential - E) * g->Rp2delta2 * exp((double)i * g->twodelta) + g->delta2p4;
}

double dfo_far(const dfo_grid* g, double position, double E)   /* Numerov.h:103-108 */
{
    const double realPosition = dfo_position(g, (long)(int)position);
    return exp(-realPosition * sqrt(2. * fabs(E)) - position * g->delta * 0.5);
}

double dfo_zero(const dfo_grid* g, double position, unsigned l)   /* Numerov.h:110-116 */
{
    const int posInd = (int)position;
    const double realPosition = dfo_position(g, posInd);
    return pow(realPosition, (double)l + 1) * exp(-position * g->delta * 0.5);
}

long dfo_max_radius_index(const dfo_grid* g, double E, long maxIndexIn)   /* Numerov.h:119-136 */
{
    size_t maxIndex = (size_t)maxIndexIn;
    double val = dfo_far(g, (double)maxIndex, E);
    if (val > 1E-200) val = (double)maxIndex;   /* no effect, kept literally (Numerov.h:121-122) */
    size_t minIndex = 1;
    while (maxIndex - minIndex > 1) {
        const size_t midIndex = (maxIndex + minIndex) / 2;
        val = dfo_far(g, (double)midIndex, E);
        if (val < 1E-200) maxIndex = midIndex;
        else              minIndex = midIndex;
    }
    return (long)maxIndex;
}

/* common prologue of the three sweeps for the non-uniform grid (Numerov.h:283-291 etc.) */
static long sweep_start(const dfo_grid* g, double E)
{
    const long steps = g->N - 1;
    double startPoint = (double)steps;
    const double m = (double)dfo_max_radius_index(g, E, steps);
    if (m < startPoint) startPoint = m;
    return (long)startPoint;
}

static const double h2p12 = 1. / 12.;   /* Numerov.h:287 */

static inline double getU(double w, double funcVal)   /* Numerov.h:510-513 */
{
    return w / (1. - h2p12 * funcVal);
}

/* ======================================================================================= */
/* Numerov sweeps                                                                          */
/* ======================================================================================= */

int dfo_count_nodes(const dfo_grid* g, const double* V, unsigned l, double E, long nodesLimit,
                    long* start, long* trip)   /* Numerov.h:272-349 */
{
    const long steps = sweep_start(g, E);
    if (start) *start = steps;
    long it = 0;
    int nodesCount;

    double position = (double)steps;
    double solution = dfo_far(g, position, E);
    double prevSol = solution;
    double funcVal = dfo_f(g, V, l, E, steps);
    double wprev = (1 - h2p12 * funcVal) * solution;

    position -= 1;
    solution = dfo_far(g, position, E);
    funcVal = dfo_f(g, V, l, E, steps - 1);
    double w = (1 - h2p12 * funcVal) * solution;

    int oldSgn = (solution > 0);
    nodesCount = 0;

    int firstClassicalReturnPoint = 0;
    for (long i = steps - 2; i > 0; --i) {
        const double wnext = 2. * w - wprev + 1. * solution * funcVal;
        ++it;
        wprev = w;
        w = wnext;

        funcVal = dfo_f(g, V, l, E, i);
        prevSol = solution;
        solution = getU(w, funcVal);

        if (fabs(solution) == INFINITY) { if (trip) *trip = it; return nodesCount; }

        const int newSgn = (solution > 0);
        if (newSgn != oldSgn) {
            ++nodesCount;
            if (nodesCount > nodesLimit) { if (trip) *trip = it; return nodesCount; }
            oldSgn = newSgn;
        }

        const double effPotential = dfo_veff(g, V, l, i);
        if (effPotential <= E) firstClassicalReturnPoint = 1;
        else if (firstClassicalReturnPoint && effPotential > E) { if (trip) *trip = it; return nodesCount; }
    }

    if (nodesCount <= nodesLimit) {
        solution = solution * (2 + 1. * funcVal) - prevSol;
        if ((solution > 0) != oldSgn) ++nodesCount;
    }
    if (trip) *trip = it;
    return nodesCount;
}

double dfo_solution_in_zero(const dfo_grid* g, const double* V, unsigned l, double E, long* start)
/* Numerov.h:351-401 */
{
    const long steps = sweep_start(g, E);
    if (start) *start = steps;

    double position = (double)steps;
    double solution = dfo_far(g, position, E);
    double prevSol = solution;
    double funcVal = dfo_f(g, V, l, E, steps);
    double wprev = (1 - h2p12 * funcVal) * solution;

    position -= 1;
    solution = dfo_far(g, position, E);
    funcVal = dfo_f(g, V, l, E, steps - 1);
    double w = (1 - h2p12 * funcVal) * solution;

    for (long i = steps - 2; i > 0; --i) {
        const double wnext = 2. * w - wprev + 1. * solution * funcVal;
        wprev = w;
        w = wnext;
        funcVal = dfo_f(g, V, l, E, i);
        prevSol = solution;
        solution = getU(w, funcVal);
    }
    solution = solution * (2 + 1. * funcVal) - prevSol;
    return solution;
}

long dfo_match(const dfo_grid* g, const double* V, unsigned l, double E, double* Psi, long* start)
/* Numerov.h:403-504 */
{
    const long highLimit = g->N;          /* steps + 1 with steps = N-1 */
    const long steps = sweep_start(g, E);
    if (start) *start = steps;

    for (long i = steps + 1; i < highLimit; ++i) Psi[i] = 0;

    const double h = (double)steps / (double)steps;    /* Numerov.h:430 (== 1) */
    const double h2 = h * h;
    const double hp12 = h2 / 12.;
    const long size = steps + 1;

    double position = (double)steps;
    double solution = dfo_far(g, position, E);
    Psi[steps] = solution;
    double funcVal = dfo_f(g, V, l, E, steps);
    double wprev = (1 - hp12 * funcVal) * solution;

    position -= h;
    Psi[steps - 1] = solution = dfo_far(g, position, E);
    funcVal = dfo_f(g, V, l, E, steps - 1);
    double w = (1 - hp12 * funcVal) * solution;

    long matchPoint = 2;
    for (long i = steps - 2; i > 0; --i) {
        const double wnext = 2. * w - wprev + h2 * solution * funcVal;
        wprev = w;
        w = wnext;
        funcVal = dfo_f(g, V, l, E, i);
        Psi[i] = solution = w / (1. - hp12 * funcVal);
        if (solution < Psi[i + 1] || fabs(solution) > 1E15) {
            matchPoint = i;
            break;
        }
    }

    Psi[0] = solution = 0;
    wprev = 0;
    position = h;
    Psi[1] = solution = dfo_zero(g, position, l);
    funcVal = dfo_f(g, V, l, E, 1);
    w = (1 - hp12 * funcVal) * solution;

    for (long i = 2; i < matchPoint; ++i) {
        const double wnext = 2. * w - wprev + h2 * solution * funcVal;
        wprev = w;
        w = wnext;
        funcVal = dfo_f(g, V, l, E, i);
        Psi[i] = solution = w / (1. - hp12 * funcVal);
    }

    w = 2. * w - wprev + h2 * solution * funcVal;
    funcVal = dfo_f(g, V, l, E, matchPoint);
    solution = w / (1. - hp12 * funcVal);

    const double factor = solution / Psi[matchPoint];
    Psi[matchPoint] = solution;
    for (long i = matchPoint + 1; i < size; ++i) Psi[i] *= factor;

    return matchPoint;
}

/* ======================================================================================= */
/* level driver                                                                            */
/* ======================================================================================= */

void dfo_locate_interval(const dfo_grid* g, const double* V, double* TopEnergy, double* BottomEnergy,
                         int L, int NumNodes, double energyErr, int* ncalls)   /* DFTAtom.cpp:566-604 */
{
    int calls = 0;
    double toe = *TopEnergy;
    double boe = *BottomEnergy;
    double deltaEnergy = toe - boe;
    while (deltaEnergy > energyErr) {
        const double E = (toe + boe) / 2;
        const int cnt = dfo_count_nodes(g, V, (unsigned)L, E, NumNodes, NULL, NULL);
        ++calls;
        if (cnt > NumNodes) toe = E; else boe = E;
        deltaEnergy = toe - boe;
    }
    *TopEnergy = toe;

    boe = *BottomEnergy;
    deltaEnergy = toe - boe;
    while (deltaEnergy > energyErr) {
        const double E = (toe + boe) / 2;
        const int cnt = dfo_count_nodes(g, V, (unsigned)L, E, NumNodes, NULL, NULL);
        ++calls;
        if (cnt < NumNodes) boe = E; else toe = E;
        deltaEnergy = toe - boe;
    }
    *BottomEnergy = toe;
    if (ncalls) *ncalls = calls;
}

void dfo_normalize_nonuniform(const dfo_grid* g, double* Psi)   /* DFTAtom.cpp:36-56 */
{
    const int n = g->N;
    double* result2 = (double*)malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        Psi[i] *= exp(i * g->delta * 0.5);
        result2[i] = Psi[i] * Psi[i];
        const double cnst = g->Rp * g->delta * exp(g->delta * i);
        result2[i] *= cnst;
    }
    const double integralForSquare = dfo_simpson38(1, result2, n);
    const double unorm = 1. / sqrt(integralForSquare);
    for (int i = 0; i < n; ++i) Psi[i] *= unorm;
    free(result2);
}

int dfo_loop_over_levels(const dfo_grid* g, const double* V, dfo_level* levels, int nlevels,
                         double* newDensity, double* Eelectronic, double* BottomEnergy, int chained,
                         const double* hints)
/* DFTAtom.cpp:493-563; returns reallyConverged.
 * chained == 1: the reference (BottomEnergy = E - 3 handed to the next level, DFTAtom.cpp:541)
 * chained == 0: every level starts from the caller's BottomEnergy (unsafe for f levels, kept for study)
 * chained == 2: level k starts from hints[k] (study: hints[k] = E_{k-1} of the PREVIOUS SCF step - 3; fragile early in the SCF)
 * chained == 3: the batched GPU mode: every level starts from max(BottomEnergy, min_i Veff_l(i)) */
{
    static const double energyErr = 1E-12;   /* DFTAtom.cpp:348 */
    int reallyConverged = 1;
    const int n = g->N;
    double* result = (double*)malloc(sizeof(double) * (size_t)n);
    const double Bottom0 = *BottomEnergy;

    for (int k = 0; k < nlevels; ++k) {
        dfo_level* level = &levels[k];
        const int NumNodes = level->n - level->l;
        double TopEnergy = 50;
        if (chained == 0) *BottomEnergy = Bottom0;
        else if (chained == 2) *BottomEnergy = hints[k];
        else if (chained == 3) {
            /* batched GPU mode: un-chained, but never below the minimum of the effective potential of this l --
             * no eigenvalue lies below it, and below it the node count of l >= 1 misfires */
            double vmin = INFINITY;
            for (long i = 1; i < n; ++i) {
                const double ve = dfo_veff(g, V, (unsigned)level->l, i);
                if (ve < vmin) vmin = ve;
            }
            *BottomEnergy = (vmin > Bottom0) ? vmin : Bottom0;
        }

        int ncalls = 0;
        dfo_locate_interval(g, V, &TopEnergy, BottomEnergy, level->l, NumNodes, energyErr, &ncalls);
        level->n_count = ncalls;
        level->top = TopEnergy;
        level->bottom = *BottomEnergy;

        double delta = dfo_solution_in_zero(g, V, (unsigned)level->l, *BottomEnergy, NULL);
        int nzero = 1;
        const int sgnBottom = delta > 0;

        int didNotConverge = 1;
        for (int i = 0; i < 500; ++i) {
            level->E = (TopEnergy + *BottomEnergy) / 2;
            delta = dfo_solution_in_zero(g, V, (unsigned)level->l, level->E, NULL);
            ++nzero;
            if ((delta > 0) == sgnBottom) *BottomEnergy = level->E;
            else                          TopEnergy = level->E;
            const double absdelta = fabs(delta);
            if (TopEnergy - *BottomEnergy < energyErr && !isnan(absdelta) && absdelta < 1E15) {
                didNotConverge = 0;
                break;
            }
        }
        level->E = *BottomEnergy;
        level->n_zero = nzero;
        level->converged = !didNotConverge;
        if (didNotConverge) reallyConverged = 0;

        *BottomEnergy = level->E - 3;

        level->matchPoint = dfo_match(g, V, (unsigned)level->l, level->E, result, NULL);
        dfo_normalize_nonuniform(g, result);

        for (int i = 0; i < n - 1; ++i)
            newDensity[i] += level->occ * result[i] * result[i];
        *Eelectronic += level->occ * level->E;
    }
    free(result);
    return reallyConverged;
}

/* Level-parallel CPU baseline (SURVEY.md section 8d ii: "one thread per level"): in the un-chained clamped mode (chained == 3,
 * what the GPU path runs) the levels are independent, so each is solved by its own call of dfo_loop_over_levels with ONE level into
 * its own density buffer (0 + x is exact), and the buffers are added in level order afterwards -- bit for bit the serial result.
 * Threads only when the library is built with -fopenmp (oracle/Makefile: libdfta_oracle_omp.so) and dfo_set_level_threads(n > 1). */
static int g_level_threads = 1;
void dfo_set_level_threads(int n) { g_level_threads = n > 1 ? n : 1; }
int dfo_get_level_threads(void)
{
#ifdef _OPENMP
    return g_level_threads;
#else
    return 1;
#endif
}

static int dfo_loop_over_levels_parallel(const dfo_grid* g, const double* V, dfo_level* levels, int nlevels,
                                         double* newDensity, double* Eelectronic, double Bottom0)
{
    const int n = g->N;
    double* all = (double*)calloc((size_t)nlevels * (size_t)n, sizeof(double));
    double* eel = (double*)calloc((size_t)nlevels, sizeof(double));
    int* cv = (int*)calloc((size_t)nlevels, sizeof(int));
#ifdef _OPENMP
#pragma omp parallel for num_threads(g_level_threads) schedule(dynamic, 1)
#endif
    for (int k = 0; k < nlevels; ++k) {
        double bot = Bottom0;
        cv[k] = dfo_loop_over_levels(g, V, &levels[k], 1, all + (size_t)k * n, &eel[k], &bot, 3, NULL);
    }
    int conv = 1;
    for (int k = 0; k < nlevels; ++k) {
        const double* t = all + (size_t)k * n;
        for (int i = 0; i < n - 1; ++i) newDensity[i] += t[i];
        *Eelectronic += eel[k];
        if (!cv[k]) conv = 0;
    }
    free(all); free(eel); free(cv);
    return conv;
}

int dfo_calculate_density(const dfo_grid* g, const double* V, dfo_level* levels, int nlevels,
                          double* density, double alpha, double* newDensity, double* Eelectronic,
                          double BottomEnergy, int chained, const double* hints)   /* DFTAtom.cpp:328-343 */
{
    const double oneMinusAlpha = 1. - alpha;
    const int conv = (chained == 3 && g_level_threads > 1)
                         ? dfo_loop_over_levels_parallel(g, V, levels, nlevels, newDensity, Eelectronic, BottomEnergy)
                         : dfo_loop_over_levels(g, V, levels, nlevels, newDensity, Eelectronic, &BottomEnergy, chained, hints);
    for (int i = 1; i < g->N; ++i) {
        const double position = g->Rp * (exp(i * g->delta) - 1.);
        newDensity[i] /= fourM_PI * position * position;
        density[i] = alpha * density[i] + oneMinusAlpha * newDensity[i];
    }
    return conv;
}

/* ======================================================================================= */
/* multigrid Poisson                                                                       */
/* ======================================================================================= */

dfo_poisson* dfo_poisson_create(int levels, double dGrid)   /* PoissonSolver.cpp:8-27 */
{
    dfo_poisson* p = (dfo_poisson*)calloc(1, sizeof(dfo_poisson));
    p->levels = levels;
    p->deltaGrid = dGrid;
    p->n = (int*)malloc(sizeof(int) * (size_t)levels);
    p->Phi = (double**)malloc(sizeof(double*) * (size_t)levels);
    p->Src = (double**)malloc(sizeof(double*) * (size_t)levels);
    p->dlev = (double*)malloc(sizeof(double) * (size_t)levels);
    int size = 3;
    for (int i = levels - 1; i >= 0; --i) {
        p->n[i] = size;
        p->Phi[i] = (double*)calloc((size_t)size, sizeof(double));
        p->Src[i] = (double*)calloc((size_t)size, sizeof(double));
        size = size * 2 - 1;
    }
    double d = dGrid;
    for (int i = 0; i < levels; ++i) { p->dlev[i] = d; d *= 2; }
    return p;
}

void dfo_poisson_destroy(dfo_poisson* p)
{
    if (!p) return;
    for (int i = 0; i < p->levels; ++i) { free(p->Phi[i]); free(p->Src[i]); }
    free(p->Phi); free(p->Src); free(p->n); free(p->dlev); free(p);
}

double dfo_gauss_seidel(dfo_poisson* p, int lvl)   /* PoissonSolver.cpp:40-64 */
{
    const double* Source = p->Src[lvl];
    double* Phi = p->Phi[lvl];
    const double d = p->dlev[lvl];
    double error2 = 0;
    const int limit = p->n[lvl] - 1;
    for (int i = 1; i < limit; ++i) {
        const double savePhi = Phi[i];
        Phi[i] = 0.5 * (Source[i] + Phi[i - 1] + Phi[i + 1] - d * (Phi[i + 1] - Phi[i - 1]) * 0.5);
        const double dif = savePhi - Phi[i];
        error2 += dif * dif;
    }
    p->n_gs++;
    return sqrt(error2);
}

double dfo_iterate_gs(dfo_poisson* p, int lvl, double errorMin, int iterno)   /* PoissonSolver.cpp:66-77 */
{
    double err = 1E10;
    for (int i = 0; i < iterno; ++i) {
        err = dfo_gauss_seidel(p, lvl);
        if (err < errorMin) break;
    }
    return err;
}

void dfo_initialize(dfo_poisson* p, double errorMin)   /* PoissonSolver.cpp:80-106 */
{
    memset(p->Phi[0], 0, sizeof(double) * (size_t)p->n[0]);
    for (int i = 1; i < p->levels; ++i) {
        const int limit = p->n[i] - 1;
        for (int q = 1; q < limit; ++q) {
            p->Src[i][q] = 4 * p->Src[i - 1][2 * q];
            p->Phi[i][q] = 0;
        }
        p->Src[i][0] = p->Src[i][limit] = 0;
        p->Phi[i][0] = p->Phi[i][limit] = 0;
    }
    const int c = p->levels - 1;
    p->Phi[c][0] = p->lowB;
    p->Phi[c][p->n[c] - 1] = p->highB;
    dfo_iterate_gs(p, c, errorMin, 15);
}

void dfo_prolong(const double* src, int nsrc, double* dst)   /* PoissonSolver.cpp:110-123 */
{
    dst[0] += src[0];
    for (int i = 1; i < nsrc; ++i) {
        const int twoi = 2 * i;
        dst[twoi] += src[i];
        dst[twoi - 1] += 0.5 * (src[i - 1] + src[i]);
    }
}

void dfo_restrict(dfo_poisson* p, int lvl)   /* PoissonSolver.cpp:126-157 */
{
    const double* Phisrc = p->Phi[lvl - 1];
    const double* Ssrc = p->Src[lvl - 1];
    double* Phidst = p->Phi[lvl];
    double* Sdst = p->Src[lvl];
    const double d = p->dlev[lvl];
    for (int i = 0; i < p->n[lvl]; ++i) Phidst[i] = 0;
    const int lim = p->n[lvl] - 1;
    for (int i = 1; i < lim; ++i) {
        const int twoi = 2 * i, m = twoi - 1, q = twoi + 1;
        Sdst[i] = 4. * (Ssrc[twoi] + Phisrc[m] - 2. * Phisrc[twoi] + Phisrc[q]) - d * (Phisrc[q] - Phisrc[m]);
    }
    Sdst[0] = Sdst[lim] = 0;
    p->n_restrict++;
}

static void ascend(dfo_poisson* p, int from, int to, double errorMin, int iterno)   /* PoissonSolver.cpp:162-171 */
{
    for (int i = from; i < to;) {
        dfo_iterate_gs(p, i, errorMin, iterno);
        dfo_restrict(p, ++i);
    }
    dfo_iterate_gs(p, to, errorMin, iterno);
}

static double descend(dfo_poisson* p, int from, int to, double errorMin, int iterno)   /* PoissonSolver.cpp:173-186 */
{
    double err = 1E10;
    for (int i = from; i > to;) {
        const int im1 = i - 1;
        dfo_prolong(p->Phi[i], p->n[i], p->Phi[im1]);
        p->n_prolong++;
        err = dfo_iterate_gs(p, im1, errorMin, iterno);
        i = im1;
    }
    return err;
}

double dfo_vcycle(dfo_poisson* p, double errorMin, int iterno)   /* PoissonSolver.h:155-159 */
{
    const int last = p->levels - 1;
    ascend(p, 0, last, errorMin, iterno);
    p->n_vcycles++;
    return descend(p, last, 0, errorMin, iterno);
}

double dfo_full_cycle(dfo_poisson* p, double errorMin, double errorMinLast)   /* PoissonSolver.h:89-124 */
{
    const int numSweeps = 3;
    const int lastLevel = p->levels - 1;
    dfo_initialize(p, errorMin);
    for (int i = p->levels - 2; i > 0; --i) {
        descend(p, lastLevel, i, errorMin, numSweeps);
        ascend(p, i, lastLevel, errorMin, numSweeps);
    }
    descend(p, lastLevel, 0, errorMinLast, numSweeps);
    double err = 0;
    for (int i = 0; i < 100; ++i) {
        err = dfo_vcycle(p, errorMinLast, numSweeps);
        if (err < errorMinLast) break;
    }
    return err;
}

double dfo_solve_poisson_nonuniform(dfo_poisson* p, int Z, double maxRadius, const double* density,
                                    double* U)   /* PoissonSolver.h:51-81 + PoissonSolver.cpp:212-223 */
{
    double* Source = p->Src[0];
    const int size = p->n[0];
    {   /* FillRNonuniformR */
        const int N = size - 1;
        const double Rp = maxRadius / (exp(N * p->deltaGrid) - 1.);
        for (int i = 0; i < size; ++i) Source[i] = Rp * (exp(i * p->deltaGrid) - 1.);
    }
    const double Rp = maxRadius / (exp(((double)size - 1.) * p->deltaGrid) - 1.);
    const double delta2grid = p->deltaGrid * p->deltaGrid;
    const double Rp2delta2 = Rp * Rp * delta2grid;
    const double twodelta = 2. * p->deltaGrid;
    const double fourM_PIRp2delta2 = fourM_PI * Rp2delta2;
    const int lim = size - 1;
    for (int i = 1; i < lim; ++i)
        Source[i] *= fourM_PIRp2delta2 * exp(i * twodelta) * density[i];

    p->lowB = 0; p->highB = Z;
    const double err = dfo_full_cycle(p, 1E-3, 1E-14);
    memcpy(U, p->Phi[0], sizeof(double) * (size_t)size);
    return err;
}

/* ======================================================================================= */
/* VWN                                                                                     */
/* ======================================================================================= */

static const double aThird = 1. / 3.;                       /* ExcCorBase.h:12 */
/* VWNExcCor.h:23-41 */
static const double AP = 0.0310907, y0P = -0.10498, bP = 3.72744, cP = 12.93532;
static const double AF = 0.01554535, y0F = -0.325, bF = 7.06042, cF = 18.0578;
static const double y0alpha = -0.0047584, balpha = 1.13107, calpha = 13.0045;
#define Y0P_    (y0P * y0P + bP * y0P + cP)
#define Y0F_    (y0F * y0F + bF * y0F + cF)
#define Y0alpha_ (y0alpha * y0alpha + balpha * y0alpha + calpha)
#define Aalpha_ (-1. / (6. * M_PI * M_PI))

static double vwnF(double y, double dify, double A, double y0, double b, double c, double Y0, double Y)
/* VWNExcCor.h:43-50 */
{
    const double Q = sqrt(4 * c - b * b);
    const double twoyb = 2. * y + b;
    const double atanQ = atan(Q / twoyb);
    return A * (log(y * y / Y) + 2. * b / Q * atanQ - b * y0 / Y0 * (log(dify * dify / Y) + 2. * (b + 2. * y0) / Q * atanQ));
}

static double vwnEcDif(double y, double dify, double A, double y0, double b, double c, double Y0, double Y)
/* VWNExcCor.h:52-55 */
{
    (void)Y0;
    return A * (c * dify - b * y0 * y) / (dify * Y);
}

static double spin_f(double zeta)   /* ExcCorBase.h:14-19 */
{
    const double mul = 1. / (2. * (pow(2., aThird) - 1.));
    return mul * (pow(1. + zeta, 4. * aThird) + pow(1. - zeta, 4. * aThird) - 2.);
}

static double spin_df(double zeta)   /* ExcCorBase.h:21-26 */
{
    const double mul = 2. / (3. * (pow(2., aThird) - 1.));
    return mul * (pow(1. + zeta, aThird) - pow(1. - zeta, aThird));
}

void dfo_vwn_vexc(const double* n, double* out, size_t sz)   /* VWNExcCor.h:73-101 */
{
    const double X1 = pow(3. / (2. * M_PI), 2. * aThird);
    const double Y0P = Y0P_;
    for (size_t i = 0; i < sz; ++i) {
        const double ro = n[i];
        if (ro < 1E-18) { out[i] = 0.; continue; }
        const double rs = pow(3. / (fourM_PI * ro), aThird);
        const double y = sqrt(rs);
        const double Y = y * y + bP * y + cP;
        const double dify = y - y0P;
        out[i] = -X1 / rs + vwnF(y, dify, AP, y0P, bP, cP, Y0P, Y) - aThird * vwnEcDif(y, dify, AP, y0P, bP, cP, Y0P, Y);
    }
}

void dfo_vwn_eexcdif(const double* n, double* out, size_t sz)   /* VWNExcCor.h:103-128 */
{
    const double X1 = 0.25 * pow(3. / (2. * M_PI), 2. * aThird);
    const double Y0P = Y0P_;
    for (size_t i = 0; i < sz; ++i) {
        const double ro = n[i];
        if (ro < 1E-18) { out[i] = 0.; continue; }
        const double rs = pow(3. / (fourM_PI * ro), aThird);
        const double y = sqrt(rs);
        const double Y = y * y + bP * y + cP;
        const double dify = y - y0P;
        out[i] = X1 / rs + aThird * vwnEcDif(y, dify, AP, y0P, bP, cP, Y0P, Y);
    }
}

void dfo_vwn_vexc_lsda(const double* na, const double* nb, double* res, double* va, double* vb, size_t sz)
/* VWNExcCor.h:134-240 */
{
    const double X1 = pow(3. / (2. * M_PI), 2. * aThird);
    const double X2 = pow(2., aThird);
    const double X12 = X1 * X2;
    const double fdd = 4. / (9. * (pow(2., aThird) - 1.));
    const double Y0P = Y0P_, Y0F = Y0F_, Y0alpha = Y0alpha_, Aalpha = Aalpha_;

    for (size_t i = 0; i < sz; ++i) {
        const double roa = na[i];
        const double rob = nb[i];
        const double n = roa + rob;
        if (n < 1E-18) { res[i] = 0.; va[i] = 0.; vb[i] = 0.; continue; }

        const double rs = pow(3. / (fourM_PI * n), aThird);
        const double rsa = pow(3. / (fourM_PI * roa), aThird);
        const double rsb = pow(3. / (fourM_PI * rob), aThird);

        const double exp_ = -X1 / rs;
        const double exf = X2 * exp_;
        const double exdif = exf - exp_;

        const double exfa = -X12 / rsa;
        const double exfb = -X12 / rsb;

        const double zeta = (roa - rob) / n;
        const double zeta3 = zeta * zeta * zeta;
        const double zeta4 = zeta3 * zeta;

        const double fval = spin_f(zeta);
        const double dfval = spin_df(zeta);

        const double y = sqrt(rs);

        const double YP = y * (y + bP) + cP;
        const double difyP = y - y0P;
        const double ecp = vwnF(y, difyP, AP, y0P, bP, cP, Y0P, YP);

        const double YF = y * (y + bF) + cF;
        const double difyF = y - y0F;
        const double ecf = vwnF(y, difyF, AF, y0F, bF, cF, Y0F, YF);

        const double YA = y * (y + balpha) + calpha;
        const double difyA = y - y0alpha;
        const double eca = vwnF(y, difyA, Aalpha, y0alpha, balpha, calpha, Y0alpha, YA);

        const double ecpd = vwnEcDif(y, difyP, AP, y0P, bP, cP, Y0P, YP);
        const double ecfd = vwnEcDif(y, difyF, AF, y0F, bF, cF, Y0F, YF);
        const double ecad = vwnEcDif(y, difyA, Aalpha, y0alpha, balpha, calpha, Y0alpha, YA);

        const double deltaecfp = ecf - ecp;
        const double beta = fdd * deltaecfp / eca - 1.;
        const double opbz4 = 1. + beta * zeta4;
        const double interp = fval / fdd * opbz4;
        const double deltaec = eca * interp;

        const double betad = fdd / eca * (ecfd - ecpd - ecad * deltaecfp / eca);
        const double interpd = fval / fdd * zeta4 * betad;

        const double deriv = aThird * (ecpd + ecad * interp + eca * interpd);
        const double dterm = eca / fdd * (4. * beta * zeta3 * fval + opbz4 * dfval);

        double r = ecp + deltaec - deriv;
        va[i] = exfa + r + (1. - zeta) * dterm;
        vb[i] = exfb + r - (1. + zeta) * dterm;
        r += (exp_ + exdif * fval);
        res[i] = r;
    }
}

void dfo_vwn_eexcdif_lsda(const double* na, const double* nb, double* res, size_t sz)
/* VWNExcCor.h:242-312 */
{
    const double X1d = 0.25 * pow(3. / (2. * M_PI), 2. * aThird);
    const double X2d = pow(2., aThird);
    const double fdd = 4. / (9. * (pow(2., aThird) - 1.));
    const double Y0P = Y0P_, Y0F = Y0F_, Y0alpha = Y0alpha_, Aalpha = Aalpha_;

    for (size_t i = 0; i < sz; ++i) {
        const double roa = na[i];
        const double rob = nb[i];
        const double n = roa + rob;
        if (n < 1E-18) { res[i] = 0.; continue; }

        const double rs = pow(3. / (fourM_PI * n), aThird);
        const double expd = X1d / rs;
        const double exfd = X2d * expd;

        const double zeta = (roa - rob) / n;
        const double zeta3 = zeta * zeta * zeta;
        const double zeta4 = zeta3 * zeta;

        const double fval = spin_f(zeta);
        const double y = sqrt(rs);

        const double YP = y * (y + bP) + cP;
        const double difyP = y - y0P;
        const double ecp = vwnF(y, difyP, AP, y0P, bP, cP, Y0P, YP);

        const double YF = y * (y + bF) + cF;
        const double difyF = y - y0F;
        const double ecf = vwnF(y, difyF, AF, y0F, bF, cF, Y0F, YF);

        const double YA = y * (y + balpha) + calpha;
        const double difyA = y - y0alpha;
        const double eca = vwnF(y, difyA, Aalpha, y0alpha, balpha, calpha, Y0alpha, YA);

        const double ecpd = vwnEcDif(y, difyP, AP, y0P, bP, cP, Y0P, YP);
        const double ecfd = vwnEcDif(y, difyF, AF, y0F, bF, cF, Y0F, YF);
        const double ecad = vwnEcDif(y, difyA, Aalpha, y0alpha, balpha, calpha, Y0alpha, YA);

        const double deltaecfp = ecf - ecp;
        const double beta = fdd * deltaecfp / eca - 1.;
        const double opbz4 = 1 + beta * zeta4;
        const double interp = fval / fdd * opbz4;

        const double betad = fdd / eca * (ecfd - ecpd - ecad * deltaecfp / eca);
        const double interpd = fval / fdd * zeta4 * betad;

        const double deriv = aThird * (ecpd + ecad * interp + eca * interpd);

        res[i] = expd + (exfd - expd) * fval + deriv;
    }
}

/* ======================================================================================= */
/* quadrature                                                                              */
/* ======================================================================================= */

double dfo_trapezoid(double delta, const double* v, int sz)   /* Integral.h:11-23 */
{
    double sum = 0.5 * (v[0] + v[sz - 1]);
    const int szm1 = sz - 1;
    for (int i = 1; i < szm1; ++i) sum += v[i];
    return sum * delta;
}

double dfo_simpson13(double delta, const double* v, int sz)   /* Integral.h:25-48 */
{
    double sum = v[0] + v[sz - 1];
    double sum4 = 0, sum2 = 0;
    const int szm1 = sz - 1;
    for (int i = 1; i < szm1; ++i) {
        sum4 += v[i++];
        if (i < szm1) sum2 += v[i];
    }
    sum += 4. * sum4 + 2. * sum2;
    const double coef = 1. / 3.;
    return sum * delta * coef;
}

double dfo_simpson38(double delta, const double* v, int sz)   /* Integral.h:50-73 */
{
    double sum = v[0] + v[sz - 1];
    double sum1 = 0, sum2 = 0;
    const int szm1 = sz - 1;
    for (int i = 1; i < szm1; ++i) {
        if (i % 3 == 0) sum2 += v[i];
        else            sum1 += v[i];
    }
    sum += 3. * sum1 + 2. * sum2;
    const double coef = 3. / 8.;
    return sum * delta * coef;
}

double dfo_boole(double delta, const double* v, int sz)   /* Integral.h:75-104 */
{
    double sum = 7. * (v[0] + v[sz - 1]);
    double sum32 = 0, sum12 = 0, sum14 = 0;
    const int szm = sz - 1;
    for (int i = 1; i < szm; ++i) {
        sum32 += v[i++];
        if (i < szm) {
            if (i % 4 == 0) sum14 += v[i];
            else            sum12 += v[i];
        }
    }
    sum += 32. * sum32 + 12. * sum12 + 14. * sum14;
    const double coef = 2. / 45.;
    return sum * delta * coef;
}

double dfo_romberg(double delta, const double* v, int sz, double err, int minSteps)   /* Integral.h:106-155 */
{
    const int numPoints = sz - 1;
    int n = numPoints;
    int cnt = 0;
    while (n) { ++cnt; n >>= 1; }

    double* Rprev = (double*)calloc((size_t)cnt, sizeof(double));
    double* Rcur = (double*)calloc((size_t)cnt, sizeof(double));
    double h = delta * numPoints;
    Rprev[0] = 0.5 * h * (v[0] + v[numPoints]);

    double result = 0;
    int returned = 0;
    n = numPoints;
    for (int i = 1; i < cnt; ++i) {
        const int oldStep = n;
        n >>= 1;
        double sum = 0;
        for (int j = n; j < numPoints; j += oldStep) sum += v[j];
        h *= 0.5;
        Rcur[0] = 0.5 * Rprev[0] + h * sum;
        double nk = 1;
        for (int m = 1; m <= i; ++m) {
            nk *= 4;
            Rcur[m] = Rcur[m - 1] + (Rcur[m - 1] - Rprev[m - 1]) / (nk - 1);
        }
        if (i >= minSteps && fabs(Rcur[i] - Rprev[i - 1]) < err) { result = Rcur[i]; returned = 1; break; }
        double* t = Rcur; Rcur = Rprev; Rprev = t;
    }
    if (!returned) result = Rprev[cnt - 1];
    free(Rprev); free(Rcur);
    return result;
}

/* ======================================================================================= */
/* Aufbau                                                                                  */
/* ======================================================================================= */

static void adjust_f_block(int* nrElectrons, int Z, int N, int L)   /* AufbauPrinciple.h:101-117 */
{
    if (3 == L) {
        if ((57 == Z || 58 == Z || 64 == Z) && 3 == N) --*nrElectrons;
        else if (4 == N) {
            if (89 == Z || 90 == Z) *nrElectrons = 0;
            else if (91 == Z || 92 == Z || 93 == Z || 96 == Z) --*nrElectrons;
        }
    } else if (103 == Z && 5 == N && 2 == L) *nrElectrons = 0;
}

static int level_less(const void* a, const void* b)   /* AufbauPrinciple.h:10-13 */
{
    const dfo_level* x = (const dfo_level*)a;
    const dfo_level* y = (const dfo_level*)b;
    if (x->n != y->n) return x->n < y->n ? -1 : 1;
    if (x->l != y->l) return x->l < y->l ? -1 : 1;
    return 0;
}

int dfo_get_subshells(int Z, dfo_level* levels)   /* AufbauPrinciple.h:36-75 */
{
    int count = 0;
    int exitLoops = 0;
    int electronCount = 0;
    for (int NplusL = 0; !exitLoops && NplusL < 10; ++NplusL)
        for (int N = 0; N <= NplusL; ++N) {
            const int L = NplusL - N;
            if (L <= N) {
                int nrElectrons = 2 * (2 * L + 1);
                adjust_f_block(&nrElectrons, Z, N, L);
                if (Z - electronCount < nrElectrons) nrElectrons = Z - electronCount;
                adjust_f_block(&nrElectrons, Z, N, L);
                if (nrElectrons > 0) {
                    electronCount += nrElectrons;
                    memset(&levels[count], 0, sizeof(dfo_level));
                    levels[count].n = N; levels[count].l = L; levels[count].occ = nrElectrons;
                    ++count;
                }
                if (electronCount == Z) { exitLoops = 1; break; }
            }
        }
    qsort(levels, (size_t)count, sizeof(dfo_level), level_less);   /* std::sort, DFTAtom.cpp:367 (keys unique) */
    return count;
}

void dfo_initialize_levels(int Z, int* nAlphaE, int* nBetaE, dfo_level* la, int* nla, dfo_level* lb, int* nlb)
/* DFTAtom.cpp:611-638 */
{
    const int n = dfo_get_subshells(Z, la);
    memcpy(lb, la, sizeof(dfo_level) * (size_t)n);
    int numAlpha = 0;
    for (int i = 0; i < n; ++i) {
        const int maxe = 2 * la[i].l + 1;
        if (la[i].occ >= maxe) {
            numAlpha += maxe;
            la[i].occ = maxe;
            lb[i].occ -= maxe;
        } else {
            numAlpha += la[i].occ;
            lb[i].occ = 0;
        }
    }
    int m = 0;
    for (int i = 0; i < n; ++i) if (lb[i].occ != 0) lb[m++] = lb[i];
    *nla = n; *nlb = m;
    *nAlphaE = numAlpha;
    *nBetaE = Z - numAlpha;
}

/* ======================================================================================= */
/* SCF                                                                                     */
/* ======================================================================================= */

static double* dvec(int n) { return (double*)calloc((size_t)n, sizeof(double)); }

dfo_scf* dfo_scf_create(int lsda, int Z, int mgLevels, double alpha, double MaxR, double deltaGrid, int chained)
/* DFTAtom.cpp:351-394 (LDA), 852-906 (LSDA) */
{
    dfo_scf* s = (dfo_scf*)calloc(1, sizeof(dfo_scf));
    s->lsda = lsda; s->Z = Z; s->mgLevels = mgLevels; s->alpha = alpha; s->MaxR = MaxR;
    s->deltaGrid = deltaGrid; s->chained = chained;
    const int N = dfo_num_nodes(mgLevels);
    dfo_grid_init(&s->g, N, deltaGrid, MaxR);
    /* DFTAtom.cpp:356 computes Rp as MaxR / (exp(NumSteps * deltaGrid) - 1.) -- same value */
    s->density = dvec(N); s->densityA = dvec(N); s->densityB = dvec(N);
    s->potA = dvec(N); s->potB = dvec(N);
    s->U = dvec(N); s->Vexc = dvec(N); s->va = dvec(N); s->vb = dvec(N); s->eexc = dvec(N);
    s->newDensity = dvec(N);
    for (int k = 0; k < 4; ++k) s->tmp[k] = dvec(N);
    s->ps = dfo_poisson_create(mgLevels, deltaGrid);

    const double volume = fourM_PI / 3. * MaxR * MaxR * MaxR;
    const double Rp = s->g.Rp;
    if (!lsda) {
        s->nla = dfo_get_subshells(Z, s->la);
        const double constDens = Z / volume;
        s->density[0] = 0;
        for (int i = 1; i < N; ++i) s->density[i] = constDens;
        dfo_solve_poisson_nonuniform(s->ps, Z, MaxR, s->density, s->U);
        dfo_vwn_vexc(s->density, s->Vexc, (size_t)N);
        s->potA[0] = 0;
        for (int i = 1; i < N; ++i) {
            const double realPos = Rp * (exp(i * deltaGrid) - 1.);
            s->potA[i] = (-Z + s->U[i]) / realPos + s->Vexc[i];
        }
    } else {
        int na, nb;
        dfo_initialize_levels(Z, &na, &nb, s->la, &s->nla, s->lb, &s->nlb);
        const double cA = na / volume;
        const double cB = nb / volume;
        s->densityA[0] = s->densityB[0] = s->density[0] = 0;
        for (int i = 1; i < N; ++i) {
            s->densityA[i] = cA;
            s->densityB[i] = cB;
            s->density[i] = cA + cB;
        }
        dfo_solve_poisson_nonuniform(s->ps, Z, MaxR, s->density, s->U);
        dfo_vwn_vexc_lsda(s->densityA, s->densityB, s->Vexc, s->va, s->vb, (size_t)N);
        s->potA[0] = 0; s->potB[0] = 0;
        for (int i = 1; i < N; ++i) {
            const double realPos = Rp * (exp(i * deltaGrid) - 1.);
            const double U = (-Z + s->U[i]) / realPos;
            s->potA[i] = U + s->va[i];
            s->potB[i] = U + s->vb[i];
        }
    }
    s->Eold = 0; s->lastTimeConverged = 0; s->step = 0; s->finished = 0;
    return s;
}

void dfo_scf_destroy(dfo_scf* s)
{
    if (!s) return;
    free(s->density); free(s->densityA); free(s->densityB); free(s->potA); free(s->potB);
    free(s->U); free(s->Vexc); free(s->va); free(s->vb); free(s->eexc); free(s->newDensity);
    for (int k = 0; k < 4; ++k) free(s->tmp[k]);
    dfo_poisson_destroy(s->ps);
    free(s);
}

int dfo_scf_step(dfo_scf* s, dfo_energies* e)   /* DFTAtom.cpp:396-484 (LDA) / 908-1009 (LSDA) */
{
    static const double totalEnergyErr = 1E-11;
    const int N = s->g.N;
    const int Z = s->Z;
    const double Rp = s->g.Rp, deltaGrid = s->deltaGrid;
    double Eelectronic = 0;
    int conv;
    double *nuclear = s->tmp[0], *exccor = s->tmp[1], *hartree = s->tmp[2], *potentiale = s->tmp[3];
    double* eexcDeriv = s->eexc;

    /* chained == 2 (hinted): first step chained as the reference, later steps start level k at E_{k-1}(previous step) - 3 */
    double hintA[32], hintB[32];
    const int mode = (s->chained == 2 && s->step == 0) ? 1 : s->chained;   /* 3 (clamped un-chained) needs no first chained step */
    if (mode == 2) {
        for (int k = 0; k < s->nla; ++k) hintA[k] = (k == 0) ? -(double)Z * Z - 1. : s->la[k - 1].E - 3;
        for (int k = 0; k < s->nlb; ++k) hintB[k] = (k == 0) ? -(double)Z * Z - 1. : s->lb[k - 1].E - 3;
    }
    if (!s->lsda) {
        memset(s->newDensity, 0, sizeof(double) * (size_t)N);
        const double BottomEnergy = -(double)Z * Z - 1.;
        conv = dfo_calculate_density(&s->g, s->potA, s->la, s->nla, s->density, s->alpha, s->newDensity,
                                     &Eelectronic, BottomEnergy, mode, hintA);
        dfo_solve_poisson_nonuniform(s->ps, Z, s->MaxR, s->density, s->U);
        dfo_vwn_vexc(s->density, s->Vexc, (size_t)N);
        dfo_vwn_eexcdif(s->density, eexcDeriv, (size_t)N);

        s->potA[0] = 0; nuclear[0] = 0; exccor[0] = 0; eexcDeriv[0] = 0; hartree[0] = 0; potentiale[0] = 0;
        for (int i = 1; i < N; ++i) {
            const double expD = exp(deltaGrid * i);
            const double position = Rp * (expD - 1.);
            const double cnst = Rp * deltaGrid * expD;
            s->potA[i] = (-Z + s->U[i]) / position + s->Vexc[i];
            const double positiondensity = position * s->density[i] * cnst;
            nuclear[i] = Z * positiondensity;
            const double position2density = position * position * s->density[i] * cnst;
            exccor[i] = position2density * s->Vexc[i];
            eexcDeriv[i] = position2density * eexcDeriv[i];
            hartree[i] = positiondensity * s->U[i];
            potentiale[i] = position2density * s->potA[i];
        }
    } else {
        memset(s->newDensity, 0, sizeof(double) * (size_t)N);
        double BottomEnergy = -(double)Z * Z - 1.;
        const int c1 = dfo_calculate_density(&s->g, s->potA, s->la, s->nla, s->densityA, s->alpha, s->newDensity,
                                             &Eelectronic, BottomEnergy, mode, hintA);
        for (int i = 0; i < N; ++i) s->newDensity[i] = 0;
        BottomEnergy = -(double)Z * Z - 1.;
        const int c2 = dfo_calculate_density(&s->g, s->potB, s->lb, s->nlb, s->densityB, s->alpha, s->newDensity,
                                             &Eelectronic, BottomEnergy, mode, hintB);
        conv = c1 && c2;
        for (int i = 1; i < N; ++i) s->density[i] = s->densityA[i] + s->densityB[i];

        dfo_solve_poisson_nonuniform(s->ps, Z, s->MaxR, s->density, s->U);
        dfo_vwn_vexc_lsda(s->densityA, s->densityB, s->Vexc, s->va, s->vb, (size_t)N);
        dfo_vwn_eexcdif_lsda(s->densityA, s->densityB, eexcDeriv, (size_t)N);

        s->potA[0] = s->potB[0] = 0;
        nuclear[0] = exccor[0] = eexcDeriv[0] = hartree[0] = potentiale[0] = 0;
        for (int i = 1; i < N; ++i) {
            const double expD = exp(deltaGrid * i);
            const double position = Rp * (expD - 1.);
            const double cnst = Rp * deltaGrid * expD;
            const double U = (-Z + s->U[i]) / position;
            s->potA[i] = U + s->va[i];
            s->potB[i] = U + s->vb[i];
            const double positioncnst = position * cnst;
            const double positiondensity = positioncnst * s->density[i];
            nuclear[i] = Z * positiondensity;
            const double position2cnst = position * positioncnst;
            const double position2density = position2cnst * s->density[i];
            const double position2densityAlpha = position2cnst * s->densityA[i];
            const double position2densityBeta = position2cnst * s->densityB[i];
            exccor[i] = position2density * s->Vexc[i];
            eexcDeriv[i] = position2density * eexcDeriv[i];
            hartree[i] = positiondensity * s->U[i];
            potentiale[i] = position2densityAlpha * s->potA[i] + position2densityBeta * s->potB[i];
        }
    }

    const double Enuclear = -fourM_PI * dfo_simpson38(1, nuclear, N);
    double Exc = fourM_PI * dfo_simpson38(1, exccor, N);
    const double eExcDif = fourM_PI * dfo_simpson38(1, eexcDeriv, N);
    Exc += eExcDif;
    const double Ehartree = -2 * M_PI * dfo_simpson38(1, hartree, N);
    const double Epotential = fourM_PI * dfo_simpson38(1, potentiale, N);
    const double Ekinetic = Eelectronic - Epotential;
    const double Etotal = Eelectronic + Ehartree + eExcDif;

    if (e) {
        e->Etotal = Etotal; e->Ekinetic = Ekinetic; e->Ecoul = -Ehartree; e->Enuclear = Enuclear; e->Exc = Exc;
        e->Eelectronic = Eelectronic; e->Ehartree = Ehartree; e->eExcDif = eExcDif; e->Epotential = Epotential;
    }
    s->step++;
    if (fabs((s->Eold - Etotal) / Etotal) < totalEnergyErr && conv && s->lastTimeConverged) {
        s->finished = 1;
        return 1;
    }
    s->Eold = Etotal;
    s->lastTimeConverged = conv;
    return 0;
}

/* ======================================================================================= */
/* uniform grid r_i = i h (NumerovFunctionRegularGrid, Numerov.h:16-70, and the            */
/* IsUniform() branches of Numerov.h:272-504; DFTAtom.cpp:21-33, 213-325;                  */
/* PoissonSolver.h:20-49, PoissonSolver.cpp:200-210)                                       */
/* ======================================================================================= */
static inline double uveff(const double* V, unsigned l, double position, long i)   /* Numerov.h:21-24 */
{
    return V[i] + l * (l + 1.) / (position * position) * 0.5;
}
static inline double ufunc(const double* V, unsigned l, double E, double position, long i)   /* Numerov.h:26-31 */
{
    return 2. * (uveff(V, l, position, i) - E);
}
static inline double ufar(double position, double E) { return exp(-position * sqrt(2. * fabs(E))); }       /* Numerov.h:33-36 */
static inline double uzero(double position, unsigned l) { return pow(position, (double)l + 1.); }           /* Numerov.h:38-41 */
static inline double umax_radius(double E) { return 200. / sqrt(2. * fabs(E)); }                            /* Numerov.h:53-56 */

int dfo_ucount_nodes(const dfo_ugrid* g, const double* V, unsigned l, double E, long nodesLimit, long* start)   /* Numerov.h:272-349 */
{
    double startPoint = g->Rmax;
    long steps = g->N - 1;
    const double h = startPoint / steps, h2 = h * h, hp12 = h2 / 12.;            /* Numerov.h:276-278 */
    { const double m = umax_radius(E); if (m < startPoint) startPoint = m; }      /* std::min(startPoint, GetMaxRadius) */
    steps = (long)(startPoint / h);
    if (start) *start = steps;

    double position = startPoint;
    double solution = ufar(position, E);
    double prevSol = solution;
    double funcVal = ufunc(V, l, E, position, steps);
    double wprev = (1 - hp12 * funcVal) * solution;
    position -= h;
    solution = ufar(position, E);
    funcVal = ufunc(V, l, E, position, steps - 1);
    double w = (1 - hp12 * funcVal) * solution;

    int oldSgn = (solution > 0);
    int nodesCount = 0;
    int firstClassicalReturnPoint = 0;
    for (long i = steps - 2; i > 0; --i) {
        const double wnext = 2. * w - wprev + h2 * solution * funcVal;
        position = h * i;
        wprev = w;
        w = wnext;
        funcVal = ufunc(V, l, E, position, i);
        prevSol = solution;
        solution = w / (1. - hp12 * funcVal);
        if (fabs(solution) == INFINITY) return nodesCount;
        const int newSgn = (solution > 0);
        if (newSgn != oldSgn) {
            ++nodesCount;
            if (nodesCount > nodesLimit) return nodesCount;
            oldSgn = newSgn;
        }
        const double effPotential = uveff(V, l, position, i);
        if (effPotential <= E) firstClassicalReturnPoint = 1;
        else if (firstClassicalReturnPoint && effPotential > E) return nodesCount;
    }
    if (nodesCount <= nodesLimit) {
        solution = solution * (2 + h2 * funcVal) - prevSol;
        if ((solution > 0) != oldSgn) ++nodesCount;
    }
    return nodesCount;
}

double dfo_usolution_in_zero(const dfo_ugrid* g, const double* V, unsigned l, double E)   /* Numerov.h:351-401 */
{
    double startPoint = g->Rmax;
    long steps = g->N - 1;
    const double h = startPoint / steps, h2 = h * h, hp12 = h2 / 12.;
    { const double m = umax_radius(E); if (m < startPoint) startPoint = m; }
    steps = (long)(startPoint / h);

    double position = startPoint;
    double solution = ufar(position, E);
    double prevSol = solution;
    double funcVal = ufunc(V, l, E, position, steps);
    double wprev = (1 - hp12 * funcVal) * solution;
    position -= h;
    solution = ufar(position, E);
    funcVal = ufunc(V, l, E, position, steps - 1);
    double w = (1 - hp12 * funcVal) * solution;
    for (long i = steps - 2; i > 0; --i) {
        const double wnext = 2. * w - wprev + h2 * solution * funcVal;
        position = h * i;
        wprev = w;
        w = wnext;
        funcVal = ufunc(V, l, E, position, i);
        prevSol = solution;
        solution = w / (1. - hp12 * funcVal);
    }
    return solution * (2 + h2 * funcVal) - prevSol;
}

long dfo_umatch(const dfo_ugrid* g, const double* V, unsigned l, double E, double* Psi)   /* Numerov.h:403-504 */
{
    double startPoint = g->Rmax;
    long steps = g->N - 1;
    const long highLimit = steps + 1;
    double h = startPoint / steps;
    { const double m = umax_radius(E); if (m < startPoint) startPoint = m; }
    steps = (long)(startPoint / h);
    for (long i = steps + 1; i < highLimit; ++i) Psi[i] = 0;
    h = startPoint / steps;                                  /* Numerov.h:430: the step is re-derived from the truncated count */
    const double h2 = h * h, hp12 = h2 / 12.;
    const long size = steps + 1;

    double position = startPoint;
    double solution = ufar(position, E);
    Psi[steps] = solution;
    double funcVal = ufunc(V, l, E, position, steps);
    double wprev = (1 - hp12 * funcVal) * solution;
    position -= h;
    Psi[steps - 1] = solution = ufar(position, E);
    funcVal = ufunc(V, l, E, position, steps - 1);
    double w = (1 - hp12 * funcVal) * solution;

    long matchPoint = 2;
    for (long i = steps - 2; i > 0; --i) {
        const double wnext = 2. * w - wprev + h2 * solution * funcVal;
        position = h * i;
        wprev = w;
        w = wnext;
        funcVal = ufunc(V, l, E, position, i);
        Psi[i] = solution = w / (1. - hp12 * funcVal);
        if (solution < Psi[i + 1] || fabs(solution) > 1E15) { matchPoint = i; break; }
    }
    position = 0;
    Psi[0] = solution = 0;
    wprev = 0;
    position += h;
    Psi[1] = solution = uzero(position, l);
    funcVal = ufunc(V, l, E, position, 1);
    w = (1 - hp12 * funcVal) * solution;
    for (long i = 2; i < matchPoint; ++i) {
        const double wnext = 2. * w - wprev + h2 * solution * funcVal;
        position = h * i;
        wprev = w;
        w = wnext;
        funcVal = ufunc(V, l, E, position, i);
        Psi[i] = solution = w / (1. - hp12 * funcVal);
    }
    w = 2. * w - wprev + h2 * solution * funcVal;
    position = h * matchPoint;
    funcVal = ufunc(V, l, E, position, matchPoint);
    solution = w / (1. - hp12 * funcVal);
    const double factor = solution / Psi[matchPoint];
    Psi[matchPoint] = solution;
    for (long i = matchPoint + 1; i < size; ++i) Psi[i] *= factor;
    return matchPoint;
}

void dfo_normalize_uniform(double* Psi, int n, double h)   /* DFTAtom.cpp:21-33 */
{
    double* result2 = (double*)malloc(sizeof(double) * (size_t)n);
    for (int i = 0; i < n; ++i) result2[i] = Psi[i] * Psi[i];
    const double integralForSquare = dfo_simpson38(h, result2, n);
    const double unorm = 1. / sqrt(integralForSquare);
    for (int i = 0; i < n; ++i) Psi[i] *= unorm;
    free(result2);
}

int dfo_uloop_over_levels(const dfo_ugrid* g, const double* V, dfo_level* levels, int nlevels, double* newDensity,
                          double* Eelectronic, double* BottomEnergy)   /* DFTAtom.cpp:213-325 */
{
    static const double energyErr = 1E-12;
    int reallyConverged = 1;
    const int n = g->N;
    double* result = (double*)malloc(sizeof(double) * (size_t)n);
    for (int k = 0; k < nlevels; ++k) {
        dfo_level* level = &levels[k];
        const int NumNodes = level->n - level->l;
        const unsigned L = (unsigned)level->l;
        double TopEnergy = 50;
        {   /* LocateInterval, DFTAtom.cpp:287-325 */
            double toe = TopEnergy, boe = *BottomEnergy;
            int calls = 0;
            while (toe - boe > energyErr) {
                const double E = (toe + boe) / 2;
                ++calls;
                if (dfo_ucount_nodes(g, V, L, E, NumNodes, NULL) > NumNodes) toe = E; else boe = E;
            }
            TopEnergy = toe;
            boe = *BottomEnergy;
            while (toe - boe > energyErr) {
                const double E = (toe + boe) / 2;
                ++calls;
                if (dfo_ucount_nodes(g, V, L, E, NumNodes, NULL) < NumNodes) boe = E; else toe = E;
            }
            *BottomEnergy = toe;
            level->n_count = calls;
        }
        level->top = TopEnergy;
        level->bottom = *BottomEnergy;
        double delta = dfo_usolution_in_zero(g, V, L, *BottomEnergy);
        int nzero = 1;
        const int sgnBottom = delta > 0;
        int didNotConverge = 1;
        for (int i = 0; i < 500; ++i) {
            level->E = (TopEnergy + *BottomEnergy) / 2;
            delta = dfo_usolution_in_zero(g, V, L, level->E);
            ++nzero;
            if ((delta > 0) == sgnBottom) *BottomEnergy = level->E; else TopEnergy = level->E;
            const double absdelta = fabs(delta);
            if (TopEnergy - *BottomEnergy < energyErr && !isnan(absdelta) && absdelta < 1E15) { didNotConverge = 0; break; }
        }
        level->E = *BottomEnergy;
        level->n_zero = nzero;
        level->converged = !didNotConverge;
        if (didNotConverge) reallyConverged = 0;
        *BottomEnergy = level->E - 3;
        level->matchPoint = dfo_umatch(g, V, L, level->E, result);
        dfo_normalize_uniform(result, n, g->h);
        for (int i = 0; i < n - 1; ++i) newDensity[i] += level->occ * result[i] * result[i];
        *Eelectronic += level->occ * level->E;
    }
    free(result);
    return reallyConverged;
}

double dfo_solve_poisson_uniform(dfo_poisson* p, int Z, double maxRadius, const double* density, double* U)
/* PoissonSolver.h:20-49 + PoissonSolver.cpp:200-210; p must have been created with deltaGrid == 0 */
{
    double* Source = p->Src[0];
    const size_t size = (size_t)p->n[0];
    {   /* FillR(Source, 0, maxRadius) */
        const size_t N = size - 1;
        const double firstR = 0;
        for (size_t i = 0; i < size; ++i) Source[i] = (firstR * (N - i) + maxRadius * i) / N;
    }
    const double delta = Source[1] - Source[0];
    const double delta2 = delta * delta;
    const double delta2fourM_PI = delta2 * fourM_PI;
    for (size_t i = 0; i < size; ++i) Source[i] *= delta2fourM_PI * density[i];
    p->lowB = 0; p->highB = Z;
    const double err = dfo_full_cycle(p, 1E-3, 1E-14);
    memcpy(U, p->Phi[0], sizeof(double) * size);
    return err;
}

/* ---- precision yardstick (tests/test_scan_precision.py) --------------------------------------------------------------------------- */
#include <math.h>
double dfo_u0_yardstick(const dfo_grid* g, const double* V, unsigned l, double E, long s, int variant)
{
    const double d = g->delta, Rp = g->Rp;
    if (variant == 1) {
        typedef long double T;
        const T dl = (T)d, Rpl = (T)Rp, c2 = (T)(Rp * Rp * d * d), c4 = (T)(d * d * 0.25), El = (T)E, ll = (T)(l * (l + 1)) * (T)0.5;
        const T sq = sqrtl((T)2 * fabsl(El));
#define DFO_F(i) ({ const T r_ = Rpl * (expl(dl * (T)(i)) - 1); T v_ = (T)V[i]; if (l > 0 && (i) > 0) v_ += ll / (r_ * r_); \
                    (T)2 * (v_ - El) * c2 * expl((T)2 * dl * (T)(i)) + c4; })
        const T rs = Rpl * (expl(dl * (T)s) - 1), rs1 = Rpl * (expl(dl * (T)(s - 1)) - 1);
        const T us = expl(-rs * sq - (T)s * dl * (T)0.5), us1 = expl(-rs1 * sq - (T)(s - 1) * dl * (T)0.5);
        T wp = (1 - DFO_F(s) / 12) * us, w = (1 - DFO_F(s - 1) / 12) * us1, u = us1, uprev = us1;
        for (long k = s - 2; k > 0; --k) {
            const T wn = 2 * w - wp + u * DFO_F(k + 1);
            wp = w; w = wn;
            uprev = u; u = w / (1 - DFO_F(k) / 12);
        }
        return (double)(u * (2 + DFO_F(1)) - uprev);
#undef DFO_F
    }
    /* double: f exactly as the sweeps of this file see it */
    const double sq = sqrt(2 * fabs(E));
    const double us = exp(-dfo_position(g, s) * sq - s * d * 0.5), us1 = exp(-dfo_position(g, s - 1) * sq - (s - 1) * d * 0.5);
    double wp = (1 - dfo_f(g, V, l, E, s) / 12) * us, w = (1 - dfo_f(g, V, l, E, s - 1) / 12) * us1;
    if (variant == 0) {
        double u = us1, uprev = us1;
        for (long k = s - 2; k > 0; --k) {
            const double wn = 2 * w - wp + u * dfo_f(g, V, l, E, k + 1);
            wp = w; w = wn;
            uprev = u; u = w / (1 - dfo_f(g, V, l, E, k) / 12);
        }
        return u * (2 + dfo_f(g, V, l, E, 1)) - uprev;
    }
    double D = w - wp;
    for (long k = s - 1; k > 1; --k) {
        const double f = dfo_f(g, V, l, E, k);
        D = D + f / (1 - f / 12) * w;
        w = w + D;
    }
    const double f1 = dfo_f(g, V, l, E, 1), f2 = dfo_f(g, V, l, E, 2);
    return (w / (1 - f1 / 12)) * (2 + f1) - (w - D) / (1 - f2 / 12);
}
