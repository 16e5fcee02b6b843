// dftatom_cli.cpp -- headless stand-in for the wxWidgets front end (DFTAtomFrame.cpp:174-199 + Options.h:48-54).
//
//   dftatom_cli Z MultigridLevels alpha MaxR deltaGrid method [chained] [--integrator=NAME]
//   dftatom_cli --ini DFTAtom.ini [chained] [--integrator=NAME] [--uniform]
//
// method: 0 = LDA, 1 = LSDA (the two the GUI reaches, DFTAtomFrame.cpp:190-195); 2 / 3 = the uniform-grid entry points
// CalculateUniformLDA / LSDA, which the GUI has commented out.  --ini reads the keys the reference persists with
// wxFileConfig (Options.cpp:42-49: Z, MultigridLevels, MaxR, deltaGrid, alpha, Method; same defaults: 36, 12, 10, 0.001,
// 0.5, 0), one `key=value` per line, an optional leading '/' and [section] lines ignored.  Values are validated like the
// options dialog does (OptionsFrame.cpp:46,152-175: Z 1..118, levels 10..20, MaxR 1..90, deltaGrid in (0, 1], alpha in [0, 1]);
// any other method value is refused (exit code 2).
// --integrator: trapezoid | simpson13 | simpson38 (default, what the reference calls) | boole | romberg (README.md:81).
// --json[=FILE]: one JSON line per SCF step (17-digit energies and eigenvalues, per-level status bits and sweep counts, rounds, V-cycles,
//                phase times) to FILE, or to stderr -- the console protocol on stdout stays the reference's.
// --sweeps=exact|tolerance, --poisson=exact|tolerance|adaptive: the opt-in tolerance modes of the device path (include/dftatom_hip.h).
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#include "DFTAtom.h"

namespace {
struct Options {                         // Options.h:48-54 with the defaults of Options.cpp:6
    int Z = 36, MultigridLevels = 12;
    double MaxR = 10., deltaGrid = 0.001, alpha = 0.5;
    int method = 0;
};

bool load_ini(const char* path, Options& o)
{
    std::ifstream f(path);
    if (!f) return false;
    std::string line;
    while (std::getline(f, line)) {
        const size_t eq = line.find('=');
        if (line.empty() || line[0] == '[' || line[0] == ';' || line[0] == '#' || eq == std::string::npos) continue;
        std::string key = line.substr(0, eq), val = line.substr(eq + 1);
        while (!key.empty() && (key.back() == ' ' || key.back() == '\t')) key.pop_back();
        if (!key.empty() && key[0] == '/') key.erase(0, 1);
        if (key == "Z") o.Z = std::atoi(val.c_str());
        else if (key == "MultigridLevels") o.MultigridLevels = std::atoi(val.c_str());
        else if (key == "MaxR") o.MaxR = std::atof(val.c_str());
        else if (key == "deltaGrid") o.deltaGrid = std::atof(val.c_str());
        else if (key == "alpha") o.alpha = std::atof(val.c_str());
        else if (key == "Method") o.method = std::atoi(val.c_str());
    }
    return true;
}

const char* validate(const Options& o, bool uniform)
{
    // the options dialog's ranges (OptionsFrame.cpp:46-50,152-171); method 2 / 3 (uniform grid) is this front end's extension
    if (o.Z < 1 || o.Z > 118) return "Z must be between 1 and 118";
    if (o.MultigridLevels < 10 || o.MultigridLevels > 20) return "Please enter between 10 and 20 levels";
    if (!(o.MaxR >= 1 && o.MaxR <= 90)) return "MaxR must be between 1 and 90";
    if (!uniform && !(o.deltaGrid > 0 && o.deltaGrid <= 1)) return "deltaGrid must be in (0, 1]";
    if (!(o.alpha >= 0 && o.alpha <= 1)) return "alpha must be in [0, 1]";
    if (o.method < 0 || o.method > 1) return "method must be 0 (LDA), 1 (LSDA), 2 (uniform LDA) or 3 (uniform LSDA)";
    return nullptr;
}
}  // namespace

int main(int argc, char** argv)
{
    Options o;
    bool uniform = false, have = false;
    int pos = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "chained") DFT::DFTAtom::levelsMode = DFTA_LEVELS_CHAINED;
        else if (a == "--uniform") uniform = true;
        else if (a == "--ini" && i + 1 < argc) {
            if (!load_ini(argv[++i], o)) { std::cerr << "cannot read " << argv[i] << std::endl; return 2; }
            have = true;
        } else if (a == "--json") {
            DFT::DFTAtom::jsonOut = &std::cerr;
        } else if (a.rfind("--json=", 0) == 0) {
            static std::ofstream jf;
            jf.open(a.substr(7));
            if (!jf) { std::cerr << "cannot write " << a.substr(7) << std::endl; return 2; }
            DFT::DFTAtom::jsonOut = &jf;
        } else if (a == "--sweeps=tolerance" || a == "--sweeps=exact") {
            DFT::DFTAtom::sweepMode = a == "--sweeps=tolerance" ? DFTA_SWEEPS_TOLERANCE : DFTA_SWEEPS_EXACT;
        } else if (a == "--poisson=tolerance" || a == "--poisson=adaptive" || a == "--poisson=exact") {
            DFT::DFTAtom::poissonMode = a == "--poisson=tolerance" ? DFTA_POISSON_TOLERANCE : (a == "--poisson=adaptive" ? DFTA_POISSON_ADAPTIVE : DFTA_POISSON_EXACT);
        } else if (a.rfind("--integrator=", 0) == 0) {
            const std::string n = a.substr(13);
            const char* names[] = {"trapezoid", "simpson13", "simpson38", "boole", "romberg"};
            int r = -1;
            for (int k = 0; k < 5; ++k) if (n == names[k]) r = k;
            if (r < 0) { std::cerr << "unknown integrator " << n << std::endl; return 2; }
            DFT::DFTAtom::integrator = r;
        } else {
            switch (pos++) {
            case 0: o.Z = std::atoi(argv[i]); break;
            case 1: o.MultigridLevels = std::atoi(argv[i]); break;
            case 2: o.alpha = std::atof(argv[i]); break;
            case 3: o.MaxR = std::atof(argv[i]); break;
            case 4: o.deltaGrid = std::atof(argv[i]); break;
            case 5: o.method = std::atoi(argv[i]); have = true; break;
            default: break;
            }
        }
    }
    if (!have) {
        std::cerr << "usage: " << argv[0] << " Z MultigridLevels alpha MaxR deltaGrid method(0 LDA, 1 LSDA, 2 uniform LDA, 3 uniform LSDA) [chained] [--integrator=NAME]\n"
                  << "       " << argv[0] << " --ini DFTAtom.ini [--uniform] [chained] [--integrator=NAME]\n";
        return 2;
    }
    if (o.method == 2 || o.method == 3) { uniform = true; o.method -= 2; }
    if (const char* msg = validate(o, uniform)) { std::cerr << "error: " << msg << std::endl; return 2; }
    try {
        if (uniform) {
            if (o.method) DFT::DFTAtom::CalculateUniformLSDA(o.Z, o.MultigridLevels, o.alpha, o.MaxR);
            else          DFT::DFTAtom::CalculateUniformLDA(o.Z, o.MultigridLevels, o.alpha, o.MaxR);
        } else {
            if (o.method) DFT::DFTAtom::CalculateNonUniformLSDA(o.Z, o.MultigridLevels, o.alpha, o.MaxR, o.deltaGrid);
            else          DFT::DFTAtom::CalculateNonUniformLDA(o.Z, o.MultigridLevels, o.alpha, o.MaxR, o.deltaGrid);
        }
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    std::cout << std::endl;
    return 0;
}
