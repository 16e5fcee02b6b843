// ref_harness.cpp -- C-callable shims over the REFERENCE's own classes (test infrastructure).
//
// This file contains no reference code.  It #includes the reference headers where they lie
// (-I/root/reference/DFTAtom) and is linked with the reference's DFTAtom.cpp/PoissonSolver.cpp
// compiled from that same place (see oracle/Makefile, target `ref`).  The result
// oracle/_ref/libdfta_ref.so is git-ignored; it is used (a) to validate oracle/dfta_oracle.c
// bit-for-bit, (b) to generate tests/golden/*.npz, (c) optionally as the timed CPU baseline
// (bench.py cpu_baseline.kind == "reference").  The product path never loads it.
//
// Access trick: the level driver (LoopOverLevels, LocateInterval, NormalizeNonUniform) is
// `private static` in DFT::DFTAtom and the multigrid internals are `protected` in
// DFT::PoissonSolver.  Access control does not change symbol names, so the harness TU sees
// them as public while the reference TUs are compiled untouched.

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>
#include <math.h>

#define private public
#define protected public
#include "PoissonSolver.h"   // first: brings <math.h> that Numerov.h relies on
#include "Numerov.h"
#include "VWNExcCor.h"
#include "ExcCor.h"
#include "Integral.h"
#include "AufbauPrinciple.h"
#include "DFTAtom.h"
#undef private
#undef protected

using NumerovNU = DFT::Numerov<DFT::NumerovFunctionNonUniformGrid>;
using NumerovU = DFT::Numerov<DFT::NumerovFunctionRegularGrid>;

namespace {
struct CoutSilencer {
    std::streambuf* old;
    std::ostringstream sink;
    CoutSilencer() : old(std::cout.rdbuf(sink.rdbuf())) {}
    ~CoutSilencer() { std::cout.rdbuf(old); }
};

struct RefNumerov {
    DFT::Potential pot;
    NumerovNU* num;
    int N;
    RefNumerov(const double* V, int n, double delta, double Rmax) : N(n)
    {
        pot.m_potentialValues.assign(V, V + n);
        num = new NumerovNU(pot, delta, Rmax, static_cast<size_t>(n));
    }
    ~RefNumerov() { delete num; }
};

struct RefNumerovU {        // uniform grid: r_i = i * MaxR / (N - 1)
    DFT::Potential pot;
    NumerovU* num;
    int N;
    double MaxR;
    RefNumerovU(const double* V, int n, double Rmax) : N(n), MaxR(Rmax)
    {
        pot.m_potentialValues.assign(V, V + n);
        num = new NumerovU(pot, 0, Rmax, static_cast<size_t>(n));
    }
    ~RefNumerovU() { delete num; }
};
}

extern "C" {

// ---- Numerov, uniform grid (Numerov.h:16-70 functor; DFTAtom.cpp:213-325 driver) -----------------
void* ref_unumerov_create(const double* V, int N, double Rmax) { return new RefNumerovU(V, N, Rmax); }
void ref_unumerov_destroy(void* h) { delete static_cast<RefNumerovU*>(h); }
int ref_ucount_nodes(void* h, unsigned l, double E, long nodesLimit)
{
    RefNumerovU* r = static_cast<RefNumerovU*>(h);
    int cnt = -1;
    r->num->SolveSchrodingerCountNodes(r->MaxR, l, E, r->N - 1, nodesLimit, cnt);
    return cnt;
}
double ref_usolution_in_zero(void* h, unsigned l, double E)
{
    RefNumerovU* r = static_cast<RefNumerovU*>(h);
    return r->num->SolveSchrodingerSolutionInZero(r->MaxR, l, E, r->N - 1);
}
long ref_umatch(void* h, unsigned l, double E, double* Psi)
{
    RefNumerovU* r = static_cast<RefNumerovU*>(h);
    long mp = -1;
    std::vector<double> res = r->num->SolveSchrodingerMatchSolutionCompletely(r->MaxR, l, E, r->N - 1, mp);
    std::memcpy(Psi, res.data(), sizeof(double) * res.size());
    return mp;
}
int ref_uloop_over_levels(void* h, int nlevels, const int* n, const int* l, const int* occ, double* Eout,
                          double* newDensity, double* Eelectronic, double* BottomEnergy)
{
    RefNumerovU* r = static_cast<RefNumerovU*>(h);
    std::vector<DFT::Subshell> levels;
    for (int i = 0; i < nlevels; ++i) levels.emplace_back(DFT::Subshell(n[i], l[i], occ[i]));
    std::vector<double> nd(newDensity, newDensity + r->N);
    bool reallyConverged = true;
    CoutSilencer quiet;
    DFT::DFTAtom::LoopOverLevels(*r->num, levels, nd, *Eelectronic, *BottomEnergy, r->N - 1, r->MaxR,
                                 r->MaxR / (r->N - 1), reallyConverged, 1E-12);
    std::memcpy(newDensity, nd.data(), sizeof(double) * nd.size());
    for (int i = 0; i < nlevels; ++i) Eout[i] = levels[i].E;
    return reallyConverged ? 1 : 0;
}
void ref_normalize_uniform(double* Psi, int N, double h)
{
    std::vector<double> v(Psi, Psi + N);
    DFT::DFTAtom::NormalizeUniform(v, h);
    std::memcpy(Psi, v.data(), sizeof(double) * v.size());
}
void ref_solve_poisson_uniform(void* p, int Z, double maxRadius, const double* density, int N, double* U)
{
    std::vector<double> d(density, density + N);
    std::vector<double> u = static_cast<DFT::PoissonSolver*>(p)->SolvePoissonUniform(Z, maxRadius, d);
    std::memcpy(U, u.data(), sizeof(double) * u.size());
}

// ---- Numerov --------------------------------------------------------------------------------
void* ref_numerov_create(const double* V, int N, double delta, double Rmax)
{
    return new RefNumerov(V, N, delta, Rmax);
}
void ref_numerov_destroy(void* h) { delete static_cast<RefNumerov*>(h); }

int ref_count_nodes(void* h, unsigned l, double E, long nodesLimit)
{
    RefNumerov* r = static_cast<RefNumerov*>(h);
    int cnt = -1;
    r->num->SolveSchrodingerCountNodes(r->N - 1, l, E, r->N - 1, nodesLimit, cnt);
    return cnt;
}
double ref_solution_in_zero(void* h, unsigned l, double E)
{
    RefNumerov* r = static_cast<RefNumerov*>(h);
    return r->num->SolveSchrodingerSolutionInZero(r->N - 1, l, E, r->N - 1);
}
long ref_match(void* h, unsigned l, double E, double* Psi)
{
    RefNumerov* r = static_cast<RefNumerov*>(h);
    long mp = -1;
    std::vector<double> res = r->num->SolveSchrodingerMatchSolutionCompletely(r->N - 1, l, E, r->N - 1, mp);
    std::memcpy(Psi, res.data(), sizeof(double) * res.size());
    return mp;
}
long ref_max_radius_index(void* h, double E)
{
    RefNumerov* r = static_cast<RefNumerov*>(h);
    return static_cast<long>(r->num->function.GetMaxRadiusIndex(E, r->N - 1, 1));
}
double ref_far(void* h, double position, double E)
{
    return static_cast<RefNumerov*>(h)->num->function.GetBoundaryValueFar(position, E);
}
double ref_zero(void* h, double position, unsigned l)
{
    return static_cast<RefNumerov*>(h)->num->function.GetBoundaryValueZero(position, l);
}
double ref_f(void* h, unsigned l, double E, long i)
{
    return static_cast<RefNumerov*>(h)->num->function(l, E, static_cast<double>(i), static_cast<size_t>(i));
}
double ref_rp(void* h) { return static_cast<RefNumerov*>(h)->num->function.GetRp(); }

// ---- level driver ----------------------------------------------------------------------------
// levels in: n[], l[], occ[] ; out: E[]. newDensity (N) accumulated. returns reallyConverged.
int ref_loop_over_levels(void* h, int nlevels, const int* n, const int* l, const int* occ, double* Eout,
                         double* newDensity, double* Eelectronic, double* BottomEnergy, double delta)
{
    RefNumerov* r = static_cast<RefNumerov*>(h);
    std::vector<DFT::Subshell> levels;
    for (int i = 0; i < nlevels; ++i) levels.emplace_back(DFT::Subshell(n[i], l[i], occ[i]));
    std::vector<double> nd(newDensity, newDensity + r->N);
    bool reallyConverged = true;
    CoutSilencer quiet;
    DFT::DFTAtom::LoopOverLevels(*r->num, levels, nd, *Eelectronic, *BottomEnergy, r->N - 1,
                                 r->num->function.GetRp(), delta, reallyConverged, 1E-12);
    std::memcpy(newDensity, nd.data(), sizeof(double) * nd.size());
    for (int i = 0; i < nlevels; ++i) Eout[i] = levels[i].E;
    return reallyConverged ? 1 : 0;
}

void ref_locate_interval(void* h, double* Top, double* Bottom, int L, int NumNodes)
{
    RefNumerov* r = static_cast<RefNumerov*>(h);
    DFT::DFTAtom::LocateInterval(*r->num, *Top, *Bottom, L, r->N - 1, NumNodes, 1E-12);
}

void ref_normalize_nonuniform(double* Psi, int N, double Rp, double delta)
{
    std::vector<double> v(Psi, Psi + N);
    DFT::DFTAtom::NormalizeNonUniform(v, Rp, delta);
    std::memcpy(Psi, v.data(), sizeof(double) * v.size());
}

// ---- Poisson -----------------------------------------------------------------------------------
void* ref_poisson_create(int levels, double dGrid) { return new DFT::PoissonSolver(levels, dGrid); }
void ref_poisson_destroy(void* p) { delete static_cast<DFT::PoissonSolver*>(p); }
int ref_poisson_level_size(void* p, int lvl) { return static_cast<int>(static_cast<DFT::PoissonSolver*>(p)->PhiLevels[lvl].size()); }
void ref_poisson_set_level(void* p, int lvl, const double* Phi, const double* Src)
{
    DFT::PoissonSolver* s = static_cast<DFT::PoissonSolver*>(p);
    if (Phi) std::copy(Phi, Phi + s->PhiLevels[lvl].size(), s->PhiLevels[lvl].begin());
    if (Src) std::copy(Src, Src + s->SourceLevels[lvl].size(), s->SourceLevels[lvl].begin());
}
void ref_poisson_get_level(void* p, int lvl, double* Phi, double* Src)
{
    DFT::PoissonSolver* s = static_cast<DFT::PoissonSolver*>(p);
    if (Phi) std::copy(s->PhiLevels[lvl].begin(), s->PhiLevels[lvl].end(), Phi);
    if (Src) std::copy(s->SourceLevels[lvl].begin(), s->SourceLevels[lvl].end(), Src);
}
double ref_gauss_seidel(void* p, int lvl) { return static_cast<DFT::PoissonSolver*>(p)->GaussSeidel(lvl); }
void ref_restrict(void* p, int lvl) { static_cast<DFT::PoissonSolver*>(p)->Restrict(lvl); }
void ref_prolong(void* p, int lvlsrc)
{
    DFT::PoissonSolver* s = static_cast<DFT::PoissonSolver*>(p);
    DFT::PoissonSolver::Prolong(s->PhiLevels[lvlsrc], s->PhiLevels[lvlsrc - 1]);
}
void ref_poisson_set_boundaries(void* p, double lo, double hi) { static_cast<DFT::PoissonSolver*>(p)->SetBoundaries(lo, hi); }
void ref_poisson_initialize(void* p, double errorMin) { static_cast<DFT::PoissonSolver*>(p)->Initialize(errorMin); }
double ref_vcycle(void* p, double errorMin, int iterno)
{
    DFT::PoissonSolver* s = static_cast<DFT::PoissonSolver*>(p);
    return s->VCycle(static_cast<int>(s->PhiLevels.size() - 1), errorMin, iterno);
}
double ref_full_cycle(void* p, double e1, double e2) { return static_cast<DFT::PoissonSolver*>(p)->FullCycle(e1, e2); }
void ref_solve_poisson_nonuniform(void* p, int Z, double maxRadius, const double* density, int N, double* U)
{
    std::vector<double> d(density, density + N);
    std::vector<double> u = static_cast<DFT::PoissonSolver*>(p)->SolvePoissonNonUniform(Z, maxRadius, d);
    std::memcpy(U, u.data(), sizeof(double) * u.size());
}
int ref_num_nodes(int levels) { return DFT::PoissonSolver::GetNumberOfNodes(levels); }

// ---- VWN ---------------------------------------------------------------------------------------
void ref_vwn_vexc(const double* n, double* out, int sz)
{
    std::vector<double> v(n, n + sz);
    std::vector<double> r = DFT::VWNExchCor::Vexc(v);
    std::memcpy(out, r.data(), sizeof(double) * r.size());
}
void ref_vwn_eexcdif(const double* n, double* out, int sz)
{
    std::vector<double> v(n, n + sz);
    std::vector<double> r = DFT::VWNExchCor::eexcDif(v);
    std::memcpy(out, r.data(), sizeof(double) * r.size());
}
void ref_vwn_vexc_lsda(const double* na, const double* nb, double* res, double* va, double* vb, int sz)
{
    std::vector<double> a(na, na + sz), b(nb, nb + sz), xa, xb;
    std::vector<double> r = DFT::VWNExchCor::Vexc(a, b, xa, xb);
    std::memcpy(res, r.data(), sizeof(double) * r.size());
    std::memcpy(va, xa.data(), sizeof(double) * xa.size());
    std::memcpy(vb, xb.data(), sizeof(double) * xb.size());
}
void ref_vwn_eexcdif_lsda(const double* na, const double* nb, double* res, int sz)
{
    std::vector<double> a(na, na + sz), b(nb, nb + sz);
    std::vector<double> r = DFT::VWNExchCor::eexcDif(a, b);
    std::memcpy(res, r.data(), sizeof(double) * r.size());
}

// ---- Chachiyo (ExcCor.h) ----------------------------------------------------------------------------
void ref_chachiyo(int improved, const double* n, double* vexc, double* eexcdif, int sz)
{
    std::vector<double> v(n, n + sz);
    std::vector<double> a = improved ? DFT::ChachiyoExchCor<DFT::ChachiyoExchCorImprovedParam>::Vexc(v)
                                     : DFT::ChachiyoExchCor<DFT::ChachiyoExchCorParam>::Vexc(v);
    std::vector<double> b = improved ? DFT::ChachiyoExchCor<DFT::ChachiyoExchCorImprovedParam>::eexcDif(v)
                                     : DFT::ChachiyoExchCor<DFT::ChachiyoExchCorParam>::eexcDif(v);
    std::memcpy(vexc, a.data(), sizeof(double) * a.size());
    std::memcpy(eexcdif, b.data(), sizeof(double) * b.size());
}

// ---- quadrature --------------------------------------------------------------------------------
double ref_integral(int which, double delta, const double* v, int sz)
{
    std::vector<double> x(v, v + sz);
    switch (which) {
    case 0: return DFT::Integral::Trapezoid(delta, x);
    case 1: return DFT::Integral::SimpsonOneThird(delta, x);
    case 2: return DFT::Integral::Simpson38(delta, x);
    case 3: return DFT::Integral::Boole(delta, x);
    default: return DFT::Integral::Romberg(delta, x);
    }
}

// ---- Aufbau ------------------------------------------------------------------------------------
int ref_get_subshells(int Z, int* n, int* l, int* occ)
{
    std::vector<DFT::Subshell> levels = DFT::AufbauPrinciple::GetSubshells(Z);
    std::sort(levels.begin(), levels.end());
    for (size_t i = 0; i < levels.size(); ++i) { n[i] = levels[i].m_N; l[i] = levels[i].m_L; occ[i] = levels[i].m_nrElectrons; }
    return static_cast<int>(levels.size());
}
void ref_initialize_levels(int Z, int* nae, int* nbe, int* na, int* an, int* al, int* aocc, int* nb, int* bn, int* bl, int* bocc)
{
    std::vector<DFT::Subshell> la, lb;
    DFT::DFTAtom::InitializeLevels(Z, *nae, *nbe, la, lb);
    *na = static_cast<int>(la.size()); *nb = static_cast<int>(lb.size());
    for (size_t i = 0; i < la.size(); ++i) { an[i] = la[i].m_N; al[i] = la[i].m_L; aocc[i] = la[i].m_nrElectrons; }
    for (size_t i = 0; i < lb.size(); ++i) { bn[i] = lb[i].m_N; bl[i] = lb[i].m_L; bocc[i] = lb[i].m_nrElectrons; }
}

// ---- end to end (stdout text of the unmodified entry points) ---------------------------------------
// mode 0 = CalculateNonUniformLDA, 1 = CalculateNonUniformLSDA.  Returns bytes written (truncated to cap-1).
long ref_calculate(int mode, int Z, int levels, double alpha, double MaxR, double delta, char* out, long cap)
{
    std::ostringstream buf;
    std::streambuf* old = std::cout.rdbuf(buf.rdbuf());
    if (mode == 1) DFT::DFTAtom::CalculateNonUniformLSDA(Z, levels, alpha, MaxR, delta);
    else           DFT::DFTAtom::CalculateNonUniformLDA(Z, levels, alpha, MaxR, delta);
    std::cout.rdbuf(old);
    const std::string s = buf.str();
    const long n = std::min<long>(static_cast<long>(s.size()), cap - 1);
    std::memcpy(out, s.data(), static_cast<size_t>(n));
    out[n] = 0;
    return n;
}

// same, from the high-precision build of the same sources (ref_hp.cpp); _steps: first max_steps SCF steps only, modes 0..3
long ref_calculate_hp(int mode, int Z, int levels, double alpha, double MaxR, double delta, char* out, long cap);
long ref_calculate_hp_steps(int mode, int Z, int levels, double alpha, double MaxR, double delta, int max_steps, char* out, long cap);

}  // extern "C"
