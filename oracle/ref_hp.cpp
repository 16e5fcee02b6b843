// ref_hp.cpp -- the reference's end-to-end entry points with 17-digit console output.
//
// Contains no reference code: it #includes the reference translation units where they lie
// (/root/reference/DFTAtom/{PoissonSolver,DFTAtom}.cpp) after (a) renaming their namespace so
// they can live next to the untouched build in one shared object and (b) widening every
// `std::setprecision(6)` in DFTAtom.cpp to 17 digits by a function-like macro.  No semantic
// change: the computation is the reference's; only the number of printed decimals differs.
// Test infrastructure only (golden-vector generation); built by `make -C oracle ref`.
#define _USE_MATH_DEFINES
#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>
#include <math.h>

#define DFT DFT_hp
#define setprecision(x) setprecision(17)
#include "PoissonSolver.cpp"
#include "DFTAtom.cpp"
#undef setprecision
#undef DFT

extern "C" long ref_calculate_hp(int mode, int Z, int levels, double alpha, double MaxR, double delta, char* out, long cap)
{
    std::ostringstream buf;
    std::streambuf* old = std::cout.rdbuf(buf.rdbuf());
    if (mode == 1) DFT_hp::DFTAtom::CalculateNonUniformLSDA(Z, levels, alpha, MaxR, delta);
    else           DFT_hp::DFTAtom::CalculateNonUniformLDA(Z, levels, alpha, MaxR, delta);
    std::cout.rdbuf(old);
    const std::string s = buf.str();
    const long n = std::min<long>(static_cast<long>(s.size()), cap - 1);
    std::memcpy(out, s.data(), static_cast<size_t>(n));
    out[n] = 0;
    return n;
}
