// DFTAtom.h -- DFT::DFTAtom with the reference's four static entry points (reference DFTAtom.h:14-18).
// The non-uniform pair runs the device-resident SCF (dfta_scf_*) and writes the reference's console text to std::cout
// (formats of DFTAtom.cpp:358,398,472,476,483,489-490,556 and 857,1015-1021), so the wxWidgets front-end
// (DFTAtomFrame.cpp:185-198) could link against it unchanged.
#pragma once

#include <vector>

#include "AufbauPrinciple.h"
#include "Numerov.h"

namespace DFT {

class DFTAtom {
public:
    static const char orb[];

    static void CalculateNonUniformLDA(int Z, int MultigridLevels, double alpha, double MaxR, double deltaGrid);
    static void CalculateUniformLDA(int Z, int MultigridLevels, double alpha, double MaxR);
    static void CalculateNonUniformLSDA(int Z, int MultigridLevels, double alpha, double MaxR, double deltaGrid);
    static void CalculateUniformLSDA(int Z, int MultigridLevels, double alpha, double MaxR);

    // knobs of the device path (not in the reference): bracket mode of the level search and maximum SCF steps
    static int levelsMode;      // DFTA_LEVELS_BATCHED (default) or DFTA_LEVELS_CHAINED (the reference's exact bisection path)

private:
    static void Run(bool lsda, int Z, int MultigridLevels, double alpha, double MaxR, double deltaGrid);
};

}  // namespace DFT
