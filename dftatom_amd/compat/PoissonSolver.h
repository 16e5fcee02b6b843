// PoissonSolver.h -- DFT::PoissonSolver with the reference's surface (reference PoissonSolver.h:15-135), executed by the
// persistent multigrid kernel through the C ABI.  Arrays are std::vector<double>, passed by const reference and
// returned by value exactly as in the reference (PoissonSolver.h:51,80).  dGrid == 0 selects the uniform grid, as in
// the reference (PoissonSolver.h:170: "for uniform, just let this be zero").
#pragma once

#define _USE_MATH_DEFINES
#include <math.h>
#include <vector>

#include "dfta_runtime.h"

namespace DFT {

class PoissonSolver {
protected:
    static constexpr double fourM_PI = 4. * M_PI;

public:
    PoissonSolver(int levels, double dGrid = 0, int Ncoarse = 3) : m_levels(levels), m_delta(dGrid)
    {
        if (Ncoarse != 3) throw std::runtime_error("PoissonSolver: only Ncoarse = 3 is supported");
    }
    ~PoissonSolver() { if (m_ps) dfta_poisson_destroy(m_ps); }
    PoissonSolver(const PoissonSolver&) = delete;
    PoissonSolver& operator=(const PoissonSolver&) = delete;

    // reference PoissonSolver.h:20-49: the grid is fixed by (levels, maxRadius)
    std::vector<double> SolvePoissonUniform(int Z, double maxRadius, const std::vector<double>& density) { return solve(true, Z, maxRadius, density); }
    // reference PoissonSolver.h:51-81; the grid is fixed by (levels, deltaGrid, maxRadius)
    std::vector<double> SolvePoissonNonUniform(int Z, double maxRadius, const std::vector<double>& density) { return solve(false, Z, maxRadius, density); }

    // reference PoissonSolver.cpp:200-223 (host loops: the solver itself works from device tables of the same values)
    static void FillR(std::vector<double>& R, double firstR, double lastR)
    {
        const size_t N = R.size() - 1;
        for (size_t i = 0; i < R.size(); ++i) R[i] = (firstR * (N - i) + lastR * i) / N;
    }
    static void FillRNonuniformR(std::vector<double>& R, double lastR, double deltaGrid)
    {
        const int N = static_cast<int>(R.size()) - 1;
        const double Rp = lastR / (exp(N * deltaGrid) - 1.);
        for (int i = 0; i <= N; ++i) R[i] = Rp * (exp(i * deltaGrid) - 1.);
    }
    // reference PoissonSolver.cpp:29-33
    void SetBoundaries(double lowBoundary, double highBoundary) { m_lowBoundary = lowBoundary; m_highBoundary = highBoundary; }
    // reference PoissonSolver.h:89-124: repeats the cycle on the source of the last SolvePoisson* call
    double FullCycle(double errorMin = 0.001, double errorMinLast = 0.00001)
    {
        auto& rt = dfta_compat::Runtime::instance();
        if (!m_ps) throw std::runtime_error("PoissonSolver::FullCycle: no source yet (call SolvePoisson* first)");
        double err = 0;
        dfta_compat::check(dfta_poisson_full_cycle(m_ps, m_lowBoundary, m_highBoundary, errorMin, errorMinLast, &err, &m_lastVcycles), rt.ctx(),
                           "dfta_poisson_full_cycle");
        m_lastErr = err;
        return err;
    }
    // the result of the last cycle: PhiLevels[0] of the reference
    std::vector<double> Solution()
    {
        auto& rt = dfta_compat::Runtime::instance();
        if (!m_ps) return {};
        std::vector<double> phi(dfta_poisson_level_size(m_ps, 0));
        dfta_compat::check(dfta_poisson_get_level(m_ps, 0, phi.data(), nullptr), rt.ctx(), "dfta_poisson_get_level");
        return phi;
    }
    static int GetNumberOfNodes(int levels, int Ncoarse = 3) { (void)Ncoarse; return dfta_num_nodes(levels); }   // PoissonSolver.h:127-135
    int lastVCycles() const { return m_lastVcycles; }
    double lastError() const { return m_lastErr; }

private:
    std::vector<double> solve(bool uniform, int Z, double maxRadius, const std::vector<double>& density)
    {
        auto& rt = dfta_compat::Runtime::instance();
        if (!m_ps || maxRadius != m_Rmax || uniform != m_uniform) {
            if (m_ps) dfta_poisson_destroy(m_ps);
            m_ps = nullptr;
            dfta_grid* g = uniform ? rt.uniform_grid(m_levels, maxRadius) : rt.grid(m_levels, m_delta, maxRadius);
            dfta_compat::check(dfta_poisson_create(rt.ctx(), g, 1, &m_ps), rt.ctx(), "dfta_poisson_create");
            m_Rmax = maxRadius;
            m_uniform = uniform;
        }
        SetBoundaries(0, Z);                                           // PoissonSolver.h:43,76
        std::vector<double> U(density.size());
        dfta_compat::check(dfta_poisson_solve(m_ps, &Z, density.data(), U.data(), &m_lastVcycles, &m_lastErr), rt.ctx(), "dfta_poisson_solve");
        return U;
    }
    int m_levels;
    double m_delta;
    double m_Rmax = -1;
    bool m_uniform = false;
    double m_lowBoundary = 0, m_highBoundary = 0;
    dfta_poisson* m_ps = nullptr;
    int m_lastVcycles = 0;
    double m_lastErr = 0;
};

}  // namespace DFT
