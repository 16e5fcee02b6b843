#!/usr/bin/env python3
"""Console protocols of the COMPILED REFERENCE (oracle/_ref/libdfta_ref.so: DFT::DFTAtom::Calculate*, six printed decimals
exactly as the wx front end shows them, DFTAtomFrame.cpp:185-198) for the front-end tests of dftatom_cli:

    python tests/golden/make_golden_cli.py      -> tests/golden/cli_protocol.json

  N_LSDA_L12   nitrogen (open 2p shell: alpha 2p3, beta 2p0 -> the beta 2p level is dropped, DFTAtom.cpp:611-638), LSDA,
               non-uniform grid, 12 levels, Rmax 15, delta 0.002: banner, per-step level / energy lines, "Finished!",
               the "Alpha:/Beta:" configuration lines (DFTAtom.cpp:1011-1021)
  Ne_LDA_L12   neon LDA on the same grid (the --ini path is compared with this capture)

The fixture is output DATA of the reference run here; nothing of its source is stored.
"""
import ctypes as C
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402


def capture(mode, Z, L, alpha, R, d):
    r = O.ref()
    buf = C.create_string_buffer(1 << 22)
    r.ref_calculate(mode, Z, L, alpha, R, d, buf, len(buf))
    return buf.value.decode()


def main():
    out = {}
    for tag, (mode, Z, L, alpha, R, d) in {"N_LSDA_L12": (1, 7, 12, 0.5, 15.0, 0.002), "Ne_LDA_L12": (0, 10, 12, 0.5, 15.0, 0.002)}.items():
        txt = capture(mode, Z, L, alpha, R, d)
        out[tag] = {"args": [Z, L, alpha, R, d, mode], "text": txt}
        print(tag, len(txt.splitlines()), "lines;", txt.strip().splitlines()[-3:])
    with open(os.path.join(HERE, "cli_protocol.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
