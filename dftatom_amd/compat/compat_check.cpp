// compat_check.cpp -- exercises the reference-shaped C++ classes the way DFTAtom.cpp uses the reference's
// (construct Potential + Numerov, call the three SolveSchrodinger* methods, PoissonSolver::SolvePoissonNonUniform,
// VWNExchCor::Vexc/eexcDif, Integral::Simpson38) and prints full-precision values, one record per line, for
// tests/test_gpu_compat.py to compare against the oracle.
#include <cmath>
#include <cstdio>
#include <vector>

#include "DFTAtom.h"
#include "Integral.h"
#include "PoissonSolver.h"
#include "VWNExcCor.h"

int main()
{
    const int L = 12;
    const double delta = 2e-3, Rmax = 25.0;
    const int N = DFT::PoissonSolver::GetNumberOfNodes(L);
    auto& rt = dfta_compat::Runtime::instance();
    std::vector<double> r(N);
    dfta_grid_get_r(rt.grid(L, delta, Rmax), r.data());

    DFT::Potential pot;
    pot.m_potentialValues.resize(N);
    pot.m_potentialValues[0] = 0;
    for (int i = 1; i < N; ++i) pot.m_potentialValues[i] = -18.0 / r[i];
    DFT::Numerov<DFT::NumerovFunctionNonUniformGrid> numerov(pot, delta, Rmax, N);
    std::printf("grid N %d Rp %.17g\n", N, numerov.function.GetRp());

    const double Es[] = {-170.0, -162.0, -100.0, -40.5, -30.0, -18.0, -5.0, -0.5};
    for (unsigned l = 0; l < 3; ++l)
        for (double E : Es) {
            int cnt = -1;
            numerov.SolveSchrodingerCountNodes(N - 1, l, E, N - 1, 3, cnt);
            const double u0 = numerov.SolveSchrodingerSolutionInZero(N - 1, l, E, N - 1);
            long mp = -1;
            const std::vector<double> psi = numerov.SolveSchrodingerMatchSolutionCompletely(N - 1, l, E, N - 1, mp);
            double s = 0;
            for (double p : psi) s += p;
            std::printf("numerov l %u E %.17g count %d u0 %.17g mp %ld psisum %.17g\n", l, E, cnt, u0, mp, s);
        }
    // the potential is re-read on every call (the reference keeps a reference to it, Numerov.h:69,186)
    for (int i = 1; i < N; ++i) pot.m_potentialValues[i] = -10.0 / r[i];
    int cnt = -1;
    numerov.SolveSchrodingerCountNodes(N - 1, 0, -20.0, N - 1, 3, cnt);
    std::printf("reread count %d\n", cnt);

    std::vector<double> rho(N);
    for (int i = 0; i < N; ++i) rho[i] = 2.0 * std::exp(-2.0 * r[i]) / M_PI;
    DFT::PoissonSolver ps(L, delta);
    const std::vector<double> U = ps.SolvePoissonNonUniform(2, Rmax, rho);
    double maxerr = 0, usum = 0;
    for (int i = 0; i < N; ++i) {
        maxerr = std::fmax(maxerr, std::fabs(U[i] - 2.0 * (1 - (1 + r[i]) * std::exp(-2 * r[i]))));
        usum += U[i];
    }
    std::printf("poisson vcycles %d maxerr %.17g usum %.17g\n", ps.lastVCycles(), maxerr, usum);

    const std::vector<double> vx = DFT::VWNExchCor::Vexc(rho), ex = DFT::VWNExchCor::eexcDif(rho);
    std::vector<double> va, vb;
    const std::vector<double> vl = DFT::VWNExchCor::Vexc(rho, rho, va, vb);
    std::printf("vwn vexc100 %.17g eexc100 %.17g lsda100 %.17g va100 %.17g mismatch_empty %d\n", vx[100], ex[100], vl[100], va[100],
                (int)DFT::VWNExchCor::Vexc(rho, std::vector<double>(3), va, vb).empty());

    std::vector<double> integrand(N);
    for (int i = 0; i < N; ++i) integrand[i] = 4 * M_PI * r[i] * r[i] * rho[i] * (numerov.function.GetRp() * delta * std::exp(delta * i));
    std::printf("integral simpson38 %.17g romberg %.17g\n", DFT::Integral::Simpson38(1.0, integrand), DFT::Integral::Romberg(1.0, integrand));

    const auto lv = DFT::AufbauPrinciple::GetSubshells(86);
    std::printf("aufbau Rn %zu first %d%c%d last %d%c%d\n", lv.size(), lv.front().m_N + 1, DFT::DFTAtom::orb[lv.front().m_L], lv.front().m_nrElectrons,
                lv.back().m_N + 1, DFT::DFTAtom::orb[lv.back().m_L], lv.back().m_nrElectrons);
    return 0;
}
