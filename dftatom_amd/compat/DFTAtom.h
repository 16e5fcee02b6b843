// DFTAtom.h -- DFT::DFTAtom as the reference declares it (reference DFTAtom.h:9-35): the four static entry points and the
// private helpers of its orchestration, so that EITHER implementation of the class links against this directory's
// L2 classes (Numerov, PoissonSolver, VWNExchCor, Integral, AufbauPrinciple):
//   * dftatom_amd/compat/DFTAtom.cpp -- the device-resident SCF (dfta_scf_*): one launch sequence per step, state in HBM;
//     it defines the four entry points (and Run / levelsMode / integrator below) and never needs the private helpers;
//   * the reference's own DFTAtom.cpp, compiled unmodified against these headers (tests/test_ref_l3.py,
//     INTEGRATION.md): its LoopOverLevels / LocateInterval / Normalize* then drive the HIP kernels call by call.
// Both write the reference's console text to std::cout, so the wxWidgets front end (DFTAtomFrame.cpp:185-198) links
// against either unchanged.
#pragma once

#include <iosfwd>
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

    // knobs of the device path (not in the reference)
    static int levelsMode;      // DFTA_LEVELS_BATCHED (default) or DFTA_LEVELS_CHAINED (the reference's exact bisection path)
    static int integrator;      // DFTA_INT_SIMPSON38 (default: what the reference calls) ... DFTA_INT_ROMBERG (what its README names)
    static int sweepMode;       // DFTA_SWEEPS_EXACT (default) / DFTA_SWEEPS_TOLERANCE (transfer-matrix scans; logarithmic grids of 12 .. 20 levels)
    static int poissonMode;     // -1 (default): as dfta_poisson_create, i.e. exact unless $DFTA_DEBUG POISSON_MODE says otherwise; DFTA_POISSON_EXACT / _TOLERANCE / _ADAPTIVE
    // per-step machine-readable output (metrics): when set, every SCF step appends ONE JSON line with 17-digit
    // energies and eigenvalues, per-level status bits (DFTA_LEVEL_*) and sweep counts, rounds, V-cycles and the phases' HIP-event times
    static std::ostream* jsonOut;

private:
    static constexpr double fourM_PI = 4. * M_PI;

    // the reference's orchestration helpers (reference DFTAtom.h:20-33); defined by the reference's DFTAtom.cpp only
    static void LoopOverLevels(Numerov<NumerovFunctionRegularGrid>& numerov, std::vector<Subshell>& levels, std::vector<double>& newDensity, double& Eelectronic, double& BottomEnergy, int NumSteps, double MaxR, double h, bool& reallyConverged, double energyErr, bool lda = true, bool isAlpha = true);
    static void LocateInterval(Numerov<NumerovFunctionRegularGrid>& numerov, double& TopEnergy, double& BottomEnergy, double MaxR, int L, int NumSteps, int NumNodes, double energyErr);
    static void CalculateNonUniformDensity(std::vector<double>& density, double alpha, double oneMinusAlpha, double deltaGrid, double Rp, int NumGridNodes, Numerov<NumerovFunctionNonUniformGrid>& numerov, std::vector<Subshell>& levels, std::vector<double>& newDensity, double& Eelectronic, double& BottomEnergy, int NumSteps, double MaxR, double h, bool& reallyConverged, double energyErr, bool lda = true, bool isAlpha = true);
    static void LoopOverLevels(Numerov<NumerovFunctionNonUniformGrid>& numerov, std::vector<Subshell>& levels, std::vector<double>& newDensity, double& Eelectronic, double& BottomEnergy, int NumSteps, double Rp, double deltaGrid, bool& reallyConverged, double energyErr, bool lda = true, bool isAlpha = true);
    static void LocateInterval(Numerov<NumerovFunctionNonUniformGrid>& numerov, double& TopEnergy, double& BottomEnergy, int L, int NumSteps, int NumNodes, double energyErr);
    static void NormalizeNonUniform(std::vector<double>& Psi, double Rp, double deltaGrid);
    static void NormalizeUniform(std::vector<double>& Psi, double h);
    static void InitializeLevels(int Z, int& numAlphaElectrons, int& numBetaElectrons, std::vector<Subshell>& levelsAlpha, std::vector<Subshell>& levelsBeta);

    // the device-resident orchestration (compat/DFTAtom.cpp)
    static void Run(bool lsda, bool uniform, int Z, int MultigridLevels, double alpha, double MaxR, double deltaGrid);
};

}  // namespace DFT
