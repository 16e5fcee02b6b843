// percall_levels.cpp -- the level search of the reference's orchestration (LocateInterval + the u(0) bisection, DFTAtom.cpp:493-604), restated
// here call by call on the PER-CALL surface of DFT::Numerov: one SolveSchrodingerCountNodes / SolveSchrodingerSolutionInZero per trial
// energy, exactly the stream of calls the reference's unmodified DFTAtom.cpp produces (tests/test_ref_l3.py links that file itself; it needs
// the reference tree, this program does not).  Prints every level's eigenvalue with 17 digits and how the calls were served: launches made
// and calls answered from what call_stream.h had integrated ahead.  tests/test_gpu_compat.py compares the eigenvalues with the oracle's
// dfo_loop_over_levels and with the run under DFTA_COMPAT_NOSPECULATE.
//     percall_levels Z L delta Rmax nlevels          e.g.  percall_levels 18 12 0.002 25 5
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "AufbauPrinciple.h"
#include "Numerov.h"
#include "PoissonSolver.h"

int main(int argc, char** argv)
{
    const int Z = argc > 1 ? atoi(argv[1]) : 18, L = argc > 2 ? atoi(argv[2]) : 12;
    const double delta = argc > 3 ? atof(argv[3]) : 2e-3, Rmax = argc > 4 ? atof(argv[4]) : 25.0;
    const size_t want = argc > 5 ? static_cast<size_t>(atoi(argv[5])) : 5;
    const int N = DFT::PoissonSolver::GetNumberOfNodes(L);
    auto& rt = dfta_compat::Runtime::instance();
    std::vector<double> r(N);
    dfta_grid_get_r(rt.grid(L, delta, Rmax), r.data());
    DFT::Potential pot;
    pot.m_potentialValues.assign(N, 0.0);
    for (int i = 1; i < N; ++i) pot.m_potentialValues[i] = -static_cast<double>(Z) / r[i];        // bare Coulomb potential
    DFT::Numerov<DFT::NumerovFunctionNonUniformGrid> numerov(pot, delta, Rmax, N);
    auto levels = DFT::AufbauPrinciple::GetSubshells(Z);
    if (levels.size() > want) levels.resize(want);

    const double err = 1E-12;
    const long steps = N - 1;
    {   // the first call pays for the upload of the potential and its tables: outside the timed part
        int c;
        numerov.SolveSchrodingerCountNodes(steps, 0, -0.5, steps, 0, c);
    }
    const auto t0 = std::chrono::steady_clock::now();
    const long launches0 = numerov.m_launches, hits0 = numerov.m_hits;
    double bottom = -static_cast<double>(Z) * Z - 1.0;                // what the reference starts an atom with (DFTAtom.cpp:407)
    long calls = 0;
    for (const auto& lv : levels) {
        const int nodes = lv.m_N - lv.m_L;
        // two node-count bisections: the band of energies with exactly `nodes` nodes
        double hi = 50, lo = bottom;
        while (hi - lo > err) {
            const double E = (hi + lo) / 2;
            int c;
            numerov.SolveSchrodingerCountNodes(steps, lv.m_L, E, steps, nodes, c); ++calls;
            if (c > nodes) hi = E; else lo = E;
        }
        const double top = hi;
        lo = bottom;
        while (hi - lo > err) {
            const double E = (hi + lo) / 2;
            int c;
            numerov.SolveSchrodingerCountNodes(steps, lv.m_L, E, steps, nodes, c); ++calls;
            if (c < nodes) lo = E; else hi = E;
        }
        // the sign change of u(0) inside it
        double B = hi, T = top;
        const bool sB = numerov.SolveSchrodingerSolutionInZero(steps, lv.m_L, B, steps) > 0; ++calls;
        int converged = 0;
        for (int i = 0; i < 500; ++i) {
            const double E = (T + B) / 2;
            const double u0 = numerov.SolveSchrodingerSolutionInZero(steps, lv.m_L, E, steps); ++calls;
            if ((u0 > 0) == sB) B = E; else T = E;
            const double a = std::fabs(u0);
            if (T - B < err && !std::isnan(a) && a < 1E15) { converged = 1; break; }
        }
        std::printf("level n %d l %d nodes %d E %.17g top %.17g converged %d\n", lv.m_N + 1, lv.m_L, nodes, B, top, converged);
        bottom = B - 3;
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("calls %ld launches %ld hits %ld seconds %.4f\n", calls, numerov.m_launches - launches0, numerov.m_hits - hits0, sec);
    return 0;
}
