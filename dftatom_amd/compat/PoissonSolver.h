// PoissonSolver.h -- DFT::PoissonSolver with the reference's surface (reference PoissonSolver.h:15-135), executed by the
// persistent multigrid kernel through the C ABI.  Arrays are std::vector<double>, passed by const reference and
// returned by value exactly as in the reference (PoissonSolver.h:51,80).
#pragma once

#define _USE_MATH_DEFINES
#include <math.h>
#include <vector>

#include "dfta_runtime.h"

namespace DFT {

class PoissonSolver {
public:
    PoissonSolver(int levels, double dGrid = 0, int Ncoarse = 3) : m_levels(levels), m_delta(dGrid)
    {
        if (Ncoarse != 3) throw std::runtime_error("PoissonSolver: only Ncoarse = 3 is supported");
    }
    ~PoissonSolver() { if (m_ps) dfta_poisson_destroy(m_ps); }
    PoissonSolver(const PoissonSolver&) = delete;
    PoissonSolver& operator=(const PoissonSolver&) = delete;

    // reference PoissonSolver.h:51-81; the grid is fixed by (levels, deltaGrid, maxRadius)
    std::vector<double> SolvePoissonNonUniform(int Z, double maxRadius, const std::vector<double>& density)
    {
        auto& rt = dfta_compat::Runtime::instance();
        if (!m_ps || maxRadius != m_Rmax) {
            if (m_ps) dfta_poisson_destroy(m_ps);
            m_ps = nullptr;
            dfta_compat::check(dfta_poisson_create(rt.ctx(), rt.grid(m_levels, m_delta, maxRadius), 1, &m_ps), rt.ctx(), "dfta_poisson_create");
            m_Rmax = maxRadius;
        }
        std::vector<double> U(density.size());
        dfta_compat::check(dfta_poisson_solve(m_ps, &Z, density.data(), U.data(), &m_lastVcycles, &m_lastErr), rt.ctx(), "dfta_poisson_solve");
        return U;
    }
    // reference PoissonSolver.h:20-49 (uniform grid): outside the accelerated hot path
    std::vector<double> SolvePoissonUniform(int, double, const std::vector<double>&)
    {
        throw std::runtime_error("SolvePoissonUniform: the uniform-grid path is not part of the HIP hot path");
    }
    static int GetNumberOfNodes(int levels, int Ncoarse = 3) { (void)Ncoarse; return dfta_num_nodes(levels); }   // PoissonSolver.h:127-135
    int lastVCycles() const { return m_lastVcycles; }
    double lastError() const { return m_lastErr; }

private:
    int m_levels;
    double m_delta;
    double m_Rmax = -1;
    dfta_poisson* m_ps = nullptr;
    int m_lastVcycles = 0;
    double m_lastErr = 0;
};

}  // namespace DFT
