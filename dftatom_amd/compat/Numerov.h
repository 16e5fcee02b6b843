// Numerov.h -- DFT::Potential, DFT::NumerovFunctionNonUniformGrid and DFT::Numerov<> with the reference's surface
// (reference Numerov.h:7-13, 73-196, 199-518), executed by the batched HIP sweep kernels through the C ABI.
//
// Semantics kept from the reference: Numerov stores a REFERENCE to the caller's Potential and re-reads it on every
// call (Numerov.h:69,186); `startPoint`/`steps` are accepted and, as in the reference's non-uniform path, only the
// grid size matters (Numerov.h:283-291).  Boundary values are evaluated on the host with libm exactly as the
// reference does (DFTA_BOUNDARY_HOST), so node counts, u(0) and Psi are bit-identical to the reference's.
// Extension (not in the reference): the *Batch methods integrate many (l, E) trials in one launch.
#pragma once

#include <vector>

#include "dfta_runtime.h"

namespace DFT {

class Potential {
public:
    inline double operator()(size_t posIndex) const { return m_potentialValues[posIndex]; }
    std::vector<double> m_potentialValues;
};

// carries the grid parameters; f(i), boundary values and the cut-off index are evaluated inside the kernels
class NumerovFunctionNonUniformGrid {
public:
    NumerovFunctionNonUniformGrid(const Potential& pot, double delta, double Rmax, size_t numPoints)
        : m_pot(pot), m_delta(delta), m_Rmax(Rmax), m_numPoints(numPoints)
    {
        m_grid = dfta_compat::Runtime::instance().grid(dfta_compat::Runtime::levels_for_nodes(numPoints), delta, Rmax);
    }
    inline double GetRp() const { return dfta_grid_rp(m_grid); }
    inline double GetDelta() const { return m_delta; }
    inline static bool IsUniform() { return false; }
    const Potential& potential() const { return m_pot; }
    dfta_grid* grid() const { return m_grid; }
    size_t numPoints() const { return m_numPoints; }

private:
    const Potential& m_pot;
    const double m_delta, m_Rmax;
    const size_t m_numPoints;
    dfta_grid* m_grid = nullptr;
};

template <class NumerovFunction> class Numerov {
public:
    Numerov(const Potential& pot, double delta = 0, double Rmax = 0, size_t numPoints = 0) : function(pot, delta, Rmax, numPoints) {}

    // reference Numerov.h:272-349
    inline void SolveSchrodingerCountNodes(double /*startPoint*/, unsigned int l, double E, long int /*steps*/, long int nodesLimit, int& nodesCount)
    {
        const int li = static_cast<int>(l), lim = static_cast<int>(nodesLimit);
        run(DFTA_SWEEP_COUNT, 1, &li, &E, &lim, &nodesCount, nullptr);
    }
    // reference Numerov.h:351-401
    inline double SolveSchrodingerSolutionInZero(double /*startPoint*/, unsigned int l, double E, long int /*steps*/)
    {
        const int li = static_cast<int>(l);
        double u0 = 0;
        run(DFTA_SWEEP_ZERO, 1, &li, &E, nullptr, nullptr, &u0);
        return u0;
    }
    // reference Numerov.h:403-504
    inline std::vector<double> SolveSchrodingerMatchSolutionCompletely(double /*startPoint*/, unsigned int l, double E, long int /*steps*/, long int& matchPoint)
    {
        auto& rt = dfta_compat::Runtime::instance();
        const std::vector<double>& V = function.potential().m_potentialValues;
        std::vector<double> Psi(V.size());
        const int li = static_cast<int>(l);
        dfta_compat::check(dfta_numerov_match(rt.ctx(), function.grid(), DFTA_BOUNDARY_HOST, 1, V.data(), 1, nullptr, &li, &E, Psi.data(), &matchPoint),
                           rt.ctx(), "dfta_numerov_match");
        return Psi;
    }

    // ---- extensions: many trials in one launch ------------------------------------------------------------------------
    inline std::vector<int> CountNodesBatch(const std::vector<int>& l, const std::vector<double>& E, const std::vector<int>& nodesLimit)
    {
        std::vector<int> counts(E.size());
        run(DFTA_SWEEP_COUNT, static_cast<int>(E.size()), l.data(), E.data(), nodesLimit.data(), counts.data(), nullptr);
        return counts;
    }
    inline std::vector<double> SolutionInZeroBatch(const std::vector<int>& l, const std::vector<double>& E)
    {
        std::vector<double> u0(E.size());
        run(DFTA_SWEEP_ZERO, static_cast<int>(E.size()), l.data(), E.data(), nullptr, nullptr, u0.data());
        return u0;
    }

    NumerovFunction function;

private:
    void run(int kind, int n, const int* l, const double* E, const int* limit, int* counts, double* u0)
    {
        auto& rt = dfta_compat::Runtime::instance();
        const std::vector<double>& V = function.potential().m_potentialValues;
        dfta_compat::check(dfta_numerov_sweeps(rt.ctx(), function.grid(), kind, DFTA_BOUNDARY_HOST, 1, V.data(), n, nullptr, l, E, limit, counts, u0,
                                               nullptr, nullptr),
                           rt.ctx(), "dfta_numerov_sweeps");
    }
};

}  // namespace DFT
