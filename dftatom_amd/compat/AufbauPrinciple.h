// AufbauPrinciple.h -- DFT::Subshell / DFT::AufbauPrinciple with the reference's surface (reference AufbauPrinciple.h:5-75).
#pragma once

#include <vector>

#include "dfta_runtime.h"

namespace DFT {

struct Subshell {
    Subshell(int N = 0, int L = 0, int nrElectrons = 0) : m_N(N), m_L(L), m_nrElectrons(nrElectrons) {}
    bool operator<(const Subshell& o) const { return m_N < o.m_N || (m_N == o.m_N && m_L < o.m_L); }
    int m_N;      // 0-based principal index (1s -> 0)
    int m_L;
    int m_nrElectrons;
    double E = 0;
};

class AufbauPrinciple {
public:
    inline static int getMaxNrAlphaElectrons(int L) { return 2 * L + 1; }
    inline static int getMaxNrElectrons(int L) { return 2 * getMaxNrAlphaElectrons(L); }
    // Madelung filling with the f-block exceptions; returned in (N, L) order (the reference sorts right after the call,
    // DFTAtom.cpp:367)
    static std::vector<Subshell> GetSubshells(int Z)
    {
        int n[32], l[32], occ[32];
        const int cnt = dfta_get_subshells(Z, n, l, occ, 32);
        if (cnt < 0) throw std::runtime_error("GetSubshells: Z out of range");
        std::vector<Subshell> levels;
        for (int i = 0; i < cnt; ++i) levels.emplace_back(Subshell(n[i], l[i], occ[i]));
        return levels;
    }
};

}  // namespace DFT
