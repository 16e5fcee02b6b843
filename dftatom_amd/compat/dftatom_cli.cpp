// dftatom_cli.cpp -- headless stand-in for the wxWidgets front-end: the six Options (Options.h:48-54) on the command line.
//   dftatom_cli Z MultigridLevels alpha MaxR deltaGrid method(0 = LDA, 1 = LSDA) [chained]
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "DFTAtom.h"

int main(int argc, char** argv)
{
    if (argc < 7) {
        std::cerr << "usage: " << argv[0] << " Z MultigridLevels alpha MaxR deltaGrid method(0 LDA, 1 LSDA) [chained]\n";
        return 2;
    }
    const int Z = std::atoi(argv[1]), levels = std::atoi(argv[2]), method = std::atoi(argv[6]);
    const double alpha = std::atof(argv[3]), MaxR = std::atof(argv[4]), delta = std::atof(argv[5]);
    if (argc > 7 && std::strcmp(argv[7], "chained") == 0) DFT::DFTAtom::levelsMode = DFTA_LEVELS_CHAINED;
    try {
        if (method == 1) DFT::DFTAtom::CalculateNonUniformLSDA(Z, levels, alpha, MaxR, delta);
        else             DFT::DFTAtom::CalculateNonUniformLDA(Z, levels, alpha, MaxR, delta);
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    std::cout << std::endl;
    return 0;
}
