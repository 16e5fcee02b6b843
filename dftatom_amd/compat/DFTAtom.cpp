// DFTAtom.cpp -- host orchestration of an SCF run on the device path, with the reference's console protocol.
//
// One dfta_scf object holds the whole state in HBM; each dfta_scf_step is one iteration of the reference's `for sp`
// loop (DFTAtom.cpp:396-484, LSDA 908-1009).  The text below reproduces what the reference prints, line for line.
#include "DFTAtom.h"

#include <algorithm>
#include <iomanip>
#include <iostream>

namespace DFT {

const char DFTAtom::orb[] = {'s', 'p', 'd', 'f'};
int DFTAtom::levelsMode = DFTA_LEVELS_BATCHED;
int DFTAtom::integrator = DFTA_INT_SIMPSON38;
int DFTAtom::sweepMode = DFTA_SWEEPS_EXACT;
int DFTAtom::poissonMode = -1;      // as dfta_poisson_create: exact unless $DFTA_DEBUG POISSON_MODE (--poisson= overrides)
std::ostream* DFTAtom::jsonOut = nullptr;

namespace {
struct LevelLine { int n, l, occ; double E; int status, n_count, n_zero; };

std::vector<LevelLine> fetch_levels(dfta_scf* scf, int spin)
{
    const int cnt = dfta_scf_num_levels(scf, 0, spin);
    std::vector<int> n(cnt), l(cnt), occ(cnt), conv(cnt);
    std::vector<double> E(cnt);
    std::vector<LevelLine> out;
    if (cnt <= 0) return out;
    std::vector<int> st(cnt), nc(cnt), nz(cnt);
    if (dfta_scf_get_levels(scf, 0, spin, n.data(), l.data(), occ.data(), E.data(), conv.data()) != DFTA_OK) throw std::runtime_error("dfta_scf_get_levels");
    if (dfta_scf_get_level_status(scf, 0, spin, st.data(), nc.data(), nz.data()) != DFTA_OK) throw std::runtime_error("dfta_scf_get_level_status");
    for (int i = 0; i < cnt; ++i) out.push_back({n[i], l[i], occ[i], E[i], st[i], nc[i], nz[i]});
    return out;
}

// one line per SCF step: full-precision values the 6-decimal console protocol cannot carry
void json_step(std::ostream& js, int sp, int Z, bool lsda, dfta_scf* scf, const dfta_energies& e, int finished, const dfta_step_stats& st)
{
    js << std::setprecision(17) << std::defaultfloat << "{\"step\": " << sp << ", \"Z\": " << Z << ", \"finished\": " << (finished ? "true" : "false")
       << ", \"Etotal\": " << e.Etotal << ", \"Ekin\": " << e.Ekinetic << ", \"Ecoul\": " << e.Ecoul << ", \"Eenuc\": " << e.Enuclear << ", \"Exc\": " << e.Exc
       << ", \"levels\": [";
    bool first = true;
    for (int spin = 0; spin < (lsda ? 2 : 1); ++spin)
        for (const auto& lv : fetch_levels(scf, spin)) {
            js << (first ? "" : ", ") << "{\"spin\": " << spin << ", \"n\": " << lv.n + 1 << ", \"l\": " << lv.l << ", \"occ\": " << lv.occ << ", \"E\": " << lv.E
               << ", \"status\": " << lv.status << ", \"count_sweeps\": " << lv.n_count << ", \"zero_sweeps\": " << lv.n_zero << "}";
            first = false;
        }
    js << "], \"rounds\": " << st.rounds << ", \"levels_layout\": " << st.levels_layout << ", \"vcycles\": " << st.vcycles
       << ", \"sweeps_reference\": " << st.sweeps_reference << ", \"sweeps_executed\": " << st.sweeps_reference_executed << ", \"sweeps_issued\": " << st.sweeps_issued
       << ", \"ms_levels\": " << st.ms_levels << ", \"ms_sweep_kernels\": " << st.ms_sweep_kernels << ", \"ms_poisson\": " << st.ms_poisson << ", \"ms_tail\": " << st.ms_tail
       << "}" << std::endl;
}

void print_configuration(std::vector<LevelLine> levels)
{
    // levels sorted by energy for the final configuration line (DFTAtom.cpp:487-490)
    std::sort(levels.begin(), levels.end(), [](const LevelLine& a, const LevelLine& b) { return a.E < b.E; });
    for (const auto& lv : levels) std::cout << lv.n + 1 << DFTAtom::orb[lv.l] << lv.occ << " ";
}
}  // namespace

void DFTAtom::Run(bool lsda, bool uniform, int Z, int MultigridLevels, double alpha, double MaxR, double deltaGrid)
{
    auto& rt = dfta_compat::Runtime::instance();
    dfta_grid* grid = uniform ? rt.uniform_grid(MultigridLevels, MaxR) : rt.grid(MultigridLevels, deltaGrid, MaxR);
    // banners of DFTAtom.cpp:70,358,658,857 (the non-uniform LDA one does say "LSD")
    std::cout << "Computing atom with Z=" << Z
              << (uniform ? (lsda ? " using LSDA with uniform grid" : " using LDA with uniform grid")
                          : (lsda ? " using LSDA with non-uniform grid" : " using LSD with non-uniform grid")) << std::endl;

    dfta_scf* scf = nullptr;
    dfta_scf_options opt = {};                    // zero = what the reference runs
    opt.struct_size = (int)sizeof(opt);
    opt.integrator = integrator; opt.functional = DFTA_XC_VWN; opt.aufbau = DFTA_AUFBAU_REFERENCE;
    opt.poisson_mode = poissonMode; opt.sweep_mode = sweepMode;
    dfta_compat::check(dfta_scf_create_ex(rt.ctx(), grid, lsda ? 1 : 0, 1, &Z, alpha, levelsMode, 0, &opt, &scf), rt.ctx(), "dfta_scf_create");
    const int maxSteps = lsda ? 150 : 100;                                  // DFTAtom.cpp:396 / 908
    for (int sp = 0; sp < maxSteps; ++sp) {
        std::cout << "Step: " << sp << std::endl;
        dfta_step_stats stats = {};
        stats.struct_size = (int)sizeof(stats);
        dfta_compat::check(dfta_scf_step(scf, jsonOut ? &stats : nullptr), rt.ctx(), "dfta_scf_step");
        for (int spin = 0; spin < (lsda ? 2 : 1); ++spin)
            for (const auto& lv : fetch_levels(scf, spin)) {
                // DFTAtom.cpp:548-556: the non-uniform path loses the alpha/beta tag, the uniform one prints it (DFTAtom.cpp:266-273,698,717)
                std::cout << "Energy ";
                if (lsda && uniform) std::cout << (spin == 0 ? "alpha " : "beta ");
                std::cout << lv.n + 1 << orb[lv.l] << ": " << std::fixed << std::setprecision(6) << lv.E << " Num nodes: " << lv.n - lv.l << std::endl;
            }
        dfta_energies e;
        int finished = 0;
        dfta_compat::check(dfta_scf_get_energies(scf, &e, &finished), rt.ctx(), "dfta_scf_get_energies");
        std::cout << "Etotal = " << std::fixed << std::setprecision(6) << e.Etotal << " Ekin = " << e.Ekinetic << " Ecoul = " << e.Ecoul
                  << " Eenuc = " << e.Enuclear << " Exc = " << e.Exc << std::endl;
        if (jsonOut) json_step(*jsonOut, sp, Z, lsda, scf, e, finished, stats);
        if (finished) {
            std::cout << std::endl << "Finished!" << std::endl << std::endl;
            break;
        }
        std::cout << "********************************************************************************" << std::endl;
    }
    if (!lsda) print_configuration(fetch_levels(scf, 0));
    else {
        std::cout << "Alpha: ";
        print_configuration(fetch_levels(scf, 0));
        std::cout << "\nBeta: ";
        print_configuration(fetch_levels(scf, 1));
    }
    dfta_scf_destroy(scf);
}

void DFTAtom::CalculateNonUniformLDA(int Z, int MultigridLevels, double alpha, double MaxR, double deltaGrid) { Run(false, false, Z, MultigridLevels, alpha, MaxR, deltaGrid); }
void DFTAtom::CalculateNonUniformLSDA(int Z, int MultigridLevels, double alpha, double MaxR, double deltaGrid) { Run(true, false, Z, MultigridLevels, alpha, MaxR, deltaGrid); }
// DFTAtom.cpp:60-210, 646-844: the same SCF on r_i = i h (unreachable from the reference's GUI, DFTAtomFrame.cpp:191,194, but part of its surface)
void DFTAtom::CalculateUniformLDA(int Z, int MultigridLevels, double alpha, double MaxR) { Run(false, true, Z, MultigridLevels, alpha, MaxR, 0); }
void DFTAtom::CalculateUniformLSDA(int Z, int MultigridLevels, double alpha, double MaxR) { Run(true, true, Z, MultigridLevels, alpha, MaxR, 0); }

}  // namespace DFT
