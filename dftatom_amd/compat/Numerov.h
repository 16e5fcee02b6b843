// Numerov.h -- DFT::Potential, the two grid functors and DFT::Numerov<> with the reference's surface
// (reference Numerov.h:7-13, 16-70, 73-196, 199-518), executed by the batched HIP sweep kernels through the C ABI.
//
// Semantics kept from the reference: Numerov stores a REFERENCE to the caller's Potential and re-reads it on every
// call (Numerov.h:69,186).  `startPoint`/`steps` must describe the whole grid, which is how DFTAtom.cpp always calls
// (N-1 / N-1 on the logarithmic grid, MaxR / N-1 on the uniform one); the cut-off is then derived per trial as in
// Numerov.h:274-291.  Boundary values are evaluated on the host with libm exactly as the reference does
// (DFTA_BOUNDARY_HOST), so node counts, u(0) and Psi are bit-identical to the reference's.
// The functors' own public methods (f(i), boundary values, cut-off) are kept for callers that use them directly; the
// kernels evaluate the same expressions from tables and never call back into them.
// Extension (not in the reference): the *Batch methods integrate many (l, E) trials in one launch.
#pragma once

#include <math.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "call_stream.h"
#include "dfta_runtime.h"

namespace DFT {

class Potential {
public:
    inline double operator()(size_t posIndex) const { return m_potentialValues[posIndex]; }
    std::vector<double> m_potentialValues;
};

// r_i = i h (reference Numerov.h:16-70); the grid handle carries h = Rmax / (numPoints - 1)
class NumerovFunctionRegularGrid {
public:
    NumerovFunctionRegularGrid(const Potential& pot, double /*delta*/, double Rmax, size_t numPoints) : m_pot(pot), m_numPoints(numPoints)
    {
        m_grid = dfta_compat::Runtime::instance().uniform_grid(dfta_compat::Runtime::levels_for_nodes(numPoints), Rmax);
    }
    inline double GetEffectivePotential(unsigned int l, double position, size_t posIndex) const { return m_pot(posIndex) + l * (l + 1.) / (position * position) * 0.5; }
    inline double operator()(unsigned int l, double E, double position, size_t posIndex) const { return 2. * (GetEffectivePotential(l, position, posIndex) - E); }
    inline static double GetBoundaryValueFar(double position, double E) { return exp(-position * sqrt(2. * fabs(E))); }
    inline static double GetBoundaryValueZero(double position, unsigned int l) { return pow(position, static_cast<double>(l) + 1.); }
    inline static double GetMaxRadius(double E, size_t /*maxIndex*/) { return 200. / sqrt(2. * fabs(E)); }
    inline static double GetMaxRadiusIndex(double E, size_t maxIndex, double stepSize) { return std::min(GetMaxRadius(E, maxIndex) / stepSize, static_cast<double>(maxIndex)); }
    inline static double GetDerivativeStep(int /*posIndex*/, double h) { return h; }
    inline static double GetWavefunctionValue(size_t /*posIndex*/, double value) { return value; }
    inline static bool IsUniform() { return true; }
    const Potential& potential() const { return m_pot; }
    dfta_grid* grid() const { return m_grid; }
    size_t numPoints() const { return m_numPoints; }

protected:
    const Potential& m_pot;
    const size_t m_numPoints;
    dfta_grid* m_grid = nullptr;
};

// r_i = Rp (exp(i delta) - 1) (reference Numerov.h:73-196)
class NumerovFunctionNonUniformGrid {
public:
    NumerovFunctionNonUniformGrid(const Potential& pot, double delta, double Rmax, size_t numPoints)
        : m_pot(pot), m_delta(delta), m_Rmax(Rmax), m_numPoints(numPoints)
    {
        m_grid = dfta_compat::Runtime::instance().grid(dfta_compat::Runtime::levels_for_nodes(numPoints), delta, Rmax);
        Rp = dfta_grid_rp(m_grid);
    }
    inline double GetEffectivePotential(unsigned int l, double /*position*/, size_t posIndex) const
    {
        const double position = GetPosition(posIndex);
        return m_pot(posIndex) + l * (l + 1.) / (position * position) * 0.5;
    }
    inline double operator()(unsigned int l, double E, double position, size_t posIndex) const
    {
        return 2. * (GetEffectivePotential(l, position, posIndex) - E) * (Rp * Rp * (m_delta * m_delta)) * exp(posIndex * (2. * m_delta)) + m_delta * m_delta * 0.25;
    }
    inline double GetBoundaryValueFar(double position, double E) const
    {
        return exp(-GetPosition(static_cast<int>(position)) * sqrt(2. * fabs(E)) - position * m_delta * 0.5);
    }
    inline double GetBoundaryValueZero(double position, unsigned int l) const
    {
        return pow(GetPosition(static_cast<int>(position)), static_cast<double>(l) + 1) * exp(-position * m_delta * 0.5);
    }
    inline double GetMaxRadiusIndex(double E, size_t maxIndex, double /*stepSize*/) const
    {
        size_t minIndex = 1;
        while (maxIndex - minIndex > 1) {
            const size_t midIndex = (maxIndex + minIndex) / 2;
            if (GetBoundaryValueFar(static_cast<double>(midIndex), E) < 1E-200) maxIndex = midIndex; else minIndex = midIndex;
        }
        return static_cast<double>(maxIndex);
    }
    inline double GetMaxRadius(double E, size_t maxIndex) const
    {
        if (GetBoundaryValueFar(static_cast<double>(maxIndex), E) > 1E-200) return GetPosition(maxIndex);
        return GetPosition(static_cast<size_t>(GetMaxRadiusIndex(E, maxIndex, 1)));
    }
    inline double GetDerivativeStep(int posIndex, double /*h*/) const { return Rp * exp(posIndex * m_delta) * (1. - exp(-m_delta)); }
    inline double GetWavefunctionValue(size_t posIndex, double value) const { return exp(static_cast<double>(posIndex) * m_delta * 0.5) * value; }
    inline double GetRp() const { return Rp; }
    inline double GetDelta() const { return m_delta; }
    inline static bool IsUniform() { return false; }
    const Potential& potential() const { return m_pot; }
    dfta_grid* grid() const { return m_grid; }
    size_t numPoints() const { return m_numPoints; }

private:
    inline double GetPosition(size_t posIndex) const { return Rp * (exp(static_cast<double>(posIndex) * m_delta) - 1.); }
    const Potential& m_pot;
    const double m_delta, m_Rmax;
    const size_t m_numPoints;
    double Rp = 0;
    dfta_grid* m_grid = nullptr;
};

template <class NumerovFunction> class Numerov {
public:
    Numerov(const Potential& pot, double delta = 0, double Rmax = 0, size_t numPoints = 0) : function(pot, delta, Rmax, numPoints) {}
    ~Numerov() { if (m_dev) dfta_potential_destroy(m_dev); }
    Numerov(const Numerov&) = delete;
    Numerov& operator=(const Numerov&) = delete;

    // reference Numerov.h:272-349
    inline void SolveSchrodingerCountNodes(double /*startPoint*/, unsigned int l, double E, long int steps, long int nodesLimit, int& nodesCount)
    {
        whole_grid(steps);
        nodesCount = one_trial(DFTA_SWEEP_COUNT, static_cast<int>(l), static_cast<int>(nodesLimit), E).count;
    }
    // reference Numerov.h:351-401
    inline double SolveSchrodingerSolutionInZero(double /*startPoint*/, unsigned int l, double E, long int steps)
    {
        whole_grid(steps);
        return one_trial(DFTA_SWEEP_ZERO, static_cast<int>(l), 0, E).u0;
    }
    // reference Numerov.h:403-504
    inline std::vector<double> SolveSchrodingerMatchSolutionCompletely(double /*startPoint*/, unsigned int l, double E, long int steps, long int& matchPoint)
    {
        whole_grid(steps);
        auto& rt = dfta_compat::Runtime::instance();
        std::vector<double> Psi(function.potential().m_potentialValues.size());
        const int li = static_cast<int>(l);
        dfta_compat::check(dfta_potential_match(resident(), 1, &li, &E, Psi.data(), &matchPoint), rt.ctx(), "dfta_potential_match");
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
    void whole_grid(long int steps) const
    {
        if (static_cast<size_t>(steps) + 1 != function.numPoints()) throw std::runtime_error("Numerov: steps must be numPoints - 1 (the whole grid)");
    }
    // The reference reads the caller's Potential afresh on every call (Numerov.h:69,186).  Here it is resident on the device with its
    // slot tables; every call compares the caller's values with the resident copy (a host memcmp) and re-uploads only what has changed.
    dfta_potential* resident()
    {
        auto& rt = dfta_compat::Runtime::instance();
        const std::vector<double>& V = function.potential().m_potentialValues;
        if (!m_dev) dfta_compat::check(dfta_potential_create(rt.ctx(), function.grid(), V.data(), &m_dev), rt.ctx(), "dfta_potential_create");
        else dfta_compat::check(dfta_potential_update(m_dev, V.data()), rt.ctx(), "dfta_potential_update");
        return m_dev;
    }
    void run(int kind, int n, const int* l, const double* E, const int* limit, int* counts, double* u0)
    {
        auto& rt = dfta_compat::Runtime::instance();
        dfta_compat::check(dfta_potential_sweeps(resident(), kind, rt.sweep_mode(function.grid()), n, l, E, limit, counts, u0, nullptr, nullptr), rt.ctx(),
                           "dfta_potential_sweeps");
    }
    // One trial of the per-call surface.  Round 5: answered from what call_stream.h had integrated ahead when the caller is the reference's
    // LoopOverLevels (the same kernels, the same potential, the bit-identical energy); otherwise the trial is integrated now -- together
    // with every energy that loop can ask for in its next calls.  $DFTA_COMPAT_NOSPECULATE: one trial per call, as in rounds 1-4.
    dfta_compat::CallStream::Value one_trial(int kind, int l, int limit, double E)
    {
        using dfta_compat::CallStream;
        auto& rt = dfta_compat::Runtime::instance();
        CallStream::Value v{0, 0.0};
        if (!rt.speculate()) {
            run(kind, 1, &l, &E, kind == DFTA_SWEEP_COUNT ? &limit : nullptr, kind == DFTA_SWEEP_COUNT ? &v.count : nullptr, kind == DFTA_SWEEP_ZERO ? &v.u0 : nullptr);
            return v;
        }
        // the reference re-reads the caller's Potential on every call: what was integrated ahead belongs to the values it had then
        const std::vector<double>& V = function.potential().m_potentialValues;
        const int mode = rt.sweep_mode(function.grid());
        if (m_seenV.size() != V.size() || m_seen_mode != mode || memcmp(m_seenV.data(), V.data(), sizeof(double) * V.size()) != 0) {
            m_seenV = V;
            m_seen_mode = mode;
            m_stream.reset();
        }
        m_stream.sync(kind, l, limit, E);
        if (!m_stream.lookup(kind, l, limit, E, v)) {
            // exact kernels: a launch of 4 095 trials (64 blocks, one per compute unit) takes what one trial takes -- 8 191 cost more on the host
            // (enumeration, cache) than the launch they save, 16 383 are slower outright; scan sweeps: one workgroup per trial, 255 fill the machine once
            const bool scan = mode == DFTA_SWEEPS_TOLERANCE;
            m_stream.plan(kind, E, scan ? 8 : 12, scan ? 255 : 4095, m_planE);
            const int n = static_cast<int>(m_planE.size());
            m_planL.assign(n, l);
            m_planLim.assign(n, limit);
            m_planCount.assign(n, 0);
            m_planU0.assign(n, 0.0);
            run(kind, n, m_planL.data(), m_planE.data(), kind == DFTA_SWEEP_COUNT ? m_planLim.data() : nullptr,
                kind == DFTA_SWEEP_COUNT ? m_planCount.data() : nullptr, kind == DFTA_SWEEP_ZERO ? m_planU0.data() : nullptr);
            m_stream.store(kind, l, limit, m_planE, kind == DFTA_SWEEP_COUNT ? m_planCount.data() : nullptr, kind == DFTA_SWEEP_ZERO ? m_planU0.data() : nullptr);
            v = CallStream::Value{m_planCount[0], m_planU0[0]};
            ++m_launches;
        } else ++m_hits;
        m_stream.advance(v);
        return v;
    }
    dfta_potential* m_dev = nullptr;
    dfta_compat::CallStream m_stream;
    std::vector<double> m_seenV, m_planE, m_planU0;
    std::vector<int> m_planL, m_planLim, m_planCount;
    int m_seen_mode = -1;

public:
    long m_launches = 0, m_hits = 0;      // diagnostics: launches made for per-call trials / calls answered from what was integrated ahead
};

}  // namespace DFT
