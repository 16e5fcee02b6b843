// VWNExcCor.h -- DFT::VWNExchCor with the reference's static surface (reference VWNExcCor.h:73,103,134,242).
#pragma once

#include <vector>

#include "dfta_runtime.h"

namespace DFT {

class VWNExchCor {
public:
    static std::vector<double> Vexc(const std::vector<double>& n)              // VWNExcCor.h:73-101
    {
        auto& rt = dfta_compat::Runtime::instance();
        std::vector<double> v(n.size());
        dfta_compat::check(dfta_vwn_lda(rt.ctx(), n.data(), n.size(), v.data(), nullptr), rt.ctx(), "dfta_vwn_lda");
        return v;
    }
    static std::vector<double> eexcDif(const std::vector<double>& n)           // VWNExcCor.h:103-128
    {
        auto& rt = dfta_compat::Runtime::instance();
        std::vector<double> e(n.size());
        dfta_compat::check(dfta_vwn_lda(rt.ctx(), n.data(), n.size(), nullptr, e.data()), rt.ctx(), "dfta_vwn_lda");
        return e;
    }
    // VWNExcCor.h:134-240: returns {} on a size mismatch, like the reference (VWNExcCor.h:137)
    static std::vector<double> Vexc(const std::vector<double>& na, const std::vector<double>& nb, std::vector<double>& va, std::vector<double>& vb)
    {
        if (na.size() != nb.size()) return {};
        auto& rt = dfta_compat::Runtime::instance();
        std::vector<double> res(na.size());
        va.resize(na.size());
        vb.resize(na.size());
        dfta_compat::check(dfta_vwn_lsda(rt.ctx(), na.data(), nb.data(), na.size(), res.data(), va.data(), vb.data(), nullptr), rt.ctx(), "dfta_vwn_lsda");
        return res;
    }
    static std::vector<double> eexcDif(const std::vector<double>& na, const std::vector<double>& nb)   // VWNExcCor.h:242-312
    {
        if (na.size() != nb.size()) return {};
        auto& rt = dfta_compat::Runtime::instance();
        std::vector<double> e(na.size());
        dfta_compat::check(dfta_vwn_lsda(rt.ctx(), na.data(), nb.data(), na.size(), nullptr, nullptr, nullptr, e.data()), rt.ctx(), "dfta_vwn_lsda");
        return e;
    }
};

}  // namespace DFT
