// Integral.h -- DFT::Integral with the reference's static templates (reference Integral.h:11-155), T = double.
#pragma once

#include <vector>

#include "dfta_runtime.h"

namespace DFT {

class Integral {
public:
    template <typename T> static T Trapezoid(const double delta, const std::vector<T>& values) { return run(DFTA_INT_TRAPEZOID, delta, values); }
    template <typename T> static T SimpsonOneThird(const double delta, const std::vector<T>& values) { return run(DFTA_INT_SIMPSON13, delta, values); }
    template <typename T> static T Simpson38(const double delta, const std::vector<T>& values) { return run(DFTA_INT_SIMPSON38, delta, values); }
    template <typename T> static T Boole(const double delta, const std::vector<T>& values) { return run(DFTA_INT_BOOLE, delta, values); }
    // reference Integral.h:106: err and minSteps are the reference's defaults (1E-18, 3); other values are not supported
    template <typename T> static T Romberg(const double delta, const std::vector<T>& values, const double err = 1E-18, const int minSteps = 3)
    {
        if (err != 1E-18 || minSteps != 3) throw std::runtime_error("Integral::Romberg: only err = 1E-18, minSteps = 3");
        return run(DFTA_INT_ROMBERG, delta, values);
    }

private:
    static double run(int rule, double delta, const std::vector<double>& values)
    {
        auto& rt = dfta_compat::Runtime::instance();
        double r = 0;
        dfta_compat::check(dfta_integrate(rt.ctx(), rule, delta, values.data(), static_cast<int>(values.size()), &r), rt.ctx(), "dfta_integrate");
        return r;
    }
};

}  // namespace DFT
