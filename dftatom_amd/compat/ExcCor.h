// ExcCor.h -- DFT::ChachiyoExchCor<Params> with the reference's static surface (reference ExcCor.h:12-95), executed by the
// pointwise HIP kernel through the C ABI.  The reference includes this header from DFTAtom.cpp:12 and keeps every call
// commented out (DFTAtom.cpp:383,412,421); the two parameter classes select the original and the improved fit.
#pragma once

#include <vector>

#include "dfta_runtime.h"

namespace DFT {

class ChachiyoExchCorParam {          // https://aip.scitation.org/doi/10.1063/1.4958669
public:
    static constexpr double b = 20.4562557;
    static constexpr double b1 = 27.4203609;
    static constexpr int improved = 0;
};

class ChachiyoExchCorImprovedParam {  // https://aip.scitation.org/doi/10.1063/1.4964758
public:
    static constexpr double b = 21.7392245;
    static constexpr double b1 = 28.3559732;
    static constexpr int improved = 1;
};

template <class Params> class ChachiyoExchCor {
public:
    static std::vector<double> Vexc(const std::vector<double>& n)          // ExcCor.h:41-68
    {
        auto& rt = dfta_compat::Runtime::instance();
        std::vector<double> v(n.size());
        dfta_compat::check(dfta_chachiyo_lda(rt.ctx(), Params::improved, n.data(), n.size(), v.data(), nullptr), rt.ctx(), "dfta_chachiyo_lda");
        return v;
    }
    static std::vector<double> eexcDif(const std::vector<double>& n)       // ExcCor.h:71-96
    {
        auto& rt = dfta_compat::Runtime::instance();
        std::vector<double> e(n.size());
        dfta_compat::check(dfta_chachiyo_lda(rt.ctx(), Params::improved, n.data(), n.size(), nullptr, e.data()), rt.ctx(), "dfta_chachiyo_lda");
        return e;
    }
};

}  // namespace DFT
