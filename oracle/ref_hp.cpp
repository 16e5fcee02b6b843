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
#include <cstdlib>
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

namespace {
// The entry points return nothing and print as they go; to capture the first SCF steps of a long run without
// waiting for convergence, the capture buffer throws when the line "Step: <max_steps>" is written and the
// exception unwinds out of Calculate* (std::cout is told to let exceptions from its buffer through).
struct StopRun {};
struct StepLimitedBuf : std::streambuf {
    std::string text, line;
    int max_steps;
    explicit StepLimitedBuf(int m) : max_steps(m) {}
    void feed(char c)
    {
        if (c == '\n') {
            if (max_steps >= 0 && line.rfind("Step: ", 0) == 0 && std::atoi(line.c_str() + 6) >= max_steps) throw StopRun();
            text += line;
            text += '\n';
            line.clear();
        } else line += c;
    }
    int_type overflow(int_type ch) override { if (ch != traits_type::eof()) feed(static_cast<char>(ch)); return ch; }
    std::streamsize xsputn(const char* s, std::streamsize n) override { for (std::streamsize i = 0; i < n; ++i) feed(s[i]); return n; }
};

void run_mode(int mode, int Z, int levels, double alpha, double MaxR, double delta)
{
    switch (mode) {
    case 1: DFT_hp::DFTAtom::CalculateNonUniformLSDA(Z, levels, alpha, MaxR, delta); break;
    case 2: DFT_hp::DFTAtom::CalculateUniformLDA(Z, levels, alpha, MaxR); break;
    case 3: DFT_hp::DFTAtom::CalculateUniformLSDA(Z, levels, alpha, MaxR); break;
    default: DFT_hp::DFTAtom::CalculateNonUniformLDA(Z, levels, alpha, MaxR, delta); break;
    }
}
}

// mode 0/1 = CalculateNonUniformLDA/LSDA, 2/3 = CalculateUniformLDA/LSDA; max_steps < 0: run to the end, otherwise stop
// when SCF step number max_steps would begin (the text then holds steps 0 .. max_steps-1)
extern "C" long ref_calculate_hp_steps(int mode, int Z, int levels, double alpha, double MaxR, double delta, int max_steps,
                                       char* out, long cap)
{
    StepLimitedBuf buf(max_steps);
    std::streambuf* old = std::cout.rdbuf(&buf);
    const std::ios_base::iostate oldex = std::cout.exceptions();
    std::cout.exceptions(std::ios_base::badbit);
    try { run_mode(mode, Z, levels, alpha, MaxR, delta); } catch (const StopRun&) {}
    std::cout.exceptions(std::ios_base::goodbit);
    std::cout.clear();
    std::cout.exceptions(oldex);
    std::cout.rdbuf(old);
    const std::string s = buf.text + buf.line;
    const long n = std::min<long>(static_cast<long>(s.size()), cap - 1);
    std::memcpy(out, s.data(), static_cast<size_t>(n));
    out[n] = 0;
    return n;
}

extern "C" long ref_calculate_hp(int mode, int Z, int levels, double alpha, double MaxR, double delta, char* out, long cap)
{
    return ref_calculate_hp_steps(mode, Z, levels, alpha, MaxR, delta, -1, out, cap);
}
