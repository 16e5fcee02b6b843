// dfta_runtime.h -- process-wide plumbing shared by the reference-shaped C++ classes in this directory.
//
// The classes DFT::Numerov / DFT::PoissonSolver / DFT::VWNExchCor / DFT::Integral / DFT::DFTAtom keep the names,
// signatures and ownership rules of aromanro/DFTAtom's public surface and forward every call
// to the C ABI of libdftatom_hip (include/dftatom_hip.h).  There is no CPU implementation behind them: without a
// usable HIP device the first call throws std::runtime_error.
#pragma once

#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>

#include "../../include/dftatom_hip.h"

namespace dfta_compat {

inline void check(int rc, dfta_ctx* ctx, const char* what)
{
    if (rc != DFTA_OK) throw std::runtime_error(std::string(what) + ": status " + std::to_string(rc) + " " + (ctx ? dfta_last_error(ctx) : ""));
}

// one context per process (the reference is single threaded: DFTAtomFrame.cpp:176-198), grids cached by parameters
class Runtime {
public:
    static Runtime& instance()
    {
        static Runtime r;
        return r;
    }
    dfta_ctx* ctx() { return m_ctx; }
    dfta_grid* grid(int levels, double delta, double Rmax)
    {
        const auto key = std::make_tuple(levels, delta, Rmax);
        auto it = m_grids.find(key);
        if (it != m_grids.end()) return it->second;
        dfta_grid* g = nullptr;
        check(dfta_grid_create(m_ctx, levels, delta, Rmax, &g), m_ctx, "dfta_grid_create");
        m_grids[key] = g;
        return g;
    }
    dfta_grid* uniform_grid(int levels, double Rmax)
    {
        const auto key = std::make_tuple(levels, 0.0, Rmax);     // delta == 0 marks the uniform grid (as PoissonSolver's dGrid does)
        auto it = m_grids.find(key);
        if (it != m_grids.end()) return it->second;
        dfta_grid* g = nullptr;
        check(dfta_grid_create_uniform(m_ctx, levels, Rmax, &g), m_ctx, "dfta_grid_create_uniform");
        m_grids[key] = g;
        return g;
    }
    // How DFT::Numerov's sweeps are integrated: the reference's rounding sequence (default) or the transfer-matrix scans
    // (DFTA_SWEEPS_TOLERANCE: $DFTA_COMPAT_SWEEPS=tolerance or set_sweep_mode; logarithmic grids of 12 .. 20 levels, else exact)
    void set_sweep_mode(int mode) { m_sweep_mode = mode; }
    // call_stream.h: the per-call trials of DFT::Numerov integrate ahead what the reference's loop asks next ($DFTA_COMPAT_NOSPECULATE: off)
    bool speculate() const { return m_speculate; }
    void set_speculate(bool on) { m_speculate = on; }
    int sweep_mode(const dfta_grid* g) const
    {
        if (m_sweep_mode != DFTA_SWEEPS_TOLERANCE || dfta_grid_is_uniform(g)) return DFTA_SWEEPS_EXACT;
        const int n = dfta_grid_num_nodes(g);
        return (n >= 4097 && n <= 1048577) ? DFTA_SWEEPS_TOLERANCE : DFTA_SWEEPS_EXACT;
    }
    // multigrid levels for a node count 2^L + 1 (PoissonSolver.h:127-135 inverted)
    static int levels_for_nodes(size_t numPoints)
    {
        for (int L = 3; L <= 24; ++L) if (static_cast<size_t>(dfta_num_nodes(L)) == numPoints) return L;
        throw std::runtime_error("node count is not 2^L + 1");
    }

private:
    Runtime()
    {
        // the structs of include/dftatom_hip.h this layer was compiled with must be those of the library it was linked to
        if (dfta_abi_version() != DFTA_ABI_VERSION)
            throw std::runtime_error("libdftatom_hip: ABI version " + std::to_string(dfta_abi_version()) + ", this compat layer was built against " + std::to_string(DFTA_ABI_VERSION));
        const int rc = dfta_ctx_create(0, nullptr, &m_ctx);
        if (rc != DFTA_OK) throw std::runtime_error("libdftatom_hip: no usable HIP device (status " + std::to_string(rc) + "); there is no CPU fallback");
        const char* e = getenv("DFTA_COMPAT_SWEEPS");
        if (e && std::string(e) == "tolerance") m_sweep_mode = DFTA_SWEEPS_TOLERANCE;
        if (getenv("DFTA_COMPAT_NOSPECULATE")) m_speculate = false;
    }
    ~Runtime()
    {
        for (auto& kv : m_grids) dfta_grid_destroy(kv.second);
        dfta_ctx_destroy(m_ctx);
    }
    dfta_ctx* m_ctx = nullptr;
    int m_sweep_mode = DFTA_SWEEPS_EXACT;
    bool m_speculate = true;
    std::map<std::tuple<int, double, double>, dfta_grid*> m_grids;
};

}  // namespace dfta_compat
