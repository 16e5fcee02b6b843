"""CPU suite: the reference's OWN orchestration layer compiles and links against this repository's L2 classes.

BASELINE.json's north_star: "Host-side C++ keeps the DFTAtom / Numerov / PoissonSolver call surface so the existing SCF
orchestration ... still drives it through a thin C-ABI HIP layer".  The proof: the reference's DFTAtom.cpp (L3: all four
Calculate* entry points, LoopOverLevels / LocateInterval for both grids, Normalize*, InitializeLevels), read from stdin so
that its quoted #includes resolve to dftatom_amd/compat first, compiles UNMODIFIED against compat's Numerov.h,
PoissonSolver.h, VWNExcCor.h, Integral.h, AufbauPrinciple.h, DFTAtom.h and links with libdftatom_hip.so
(`make -C oracle ref_l3`).  Build container only: needs /root/reference; nothing of it is copied, the outputs land in the
git-ignored oracle/_ref/.  (On the GPU the resulting binary prints, for Ar @ 12 levels, byte for byte the text of
dftatom_cli in chained mode: INTEGRATION.md.)
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/DFTAtom"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference tree (build container only)")


def test_reference_orchestration_compiles_and_links_against_compat():
    lib = os.path.join(ROOT, "dftatom_amd", "libdftatom_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "dftatom_amd", "csrc")])
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_l3_cli")
    if os.path.exists(exe):
        os.remove(exe)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref_l3"], stdout=subprocess.DEVNULL)
    assert os.path.exists(exe)
    # which headers did the reference's translation unit see?  every L2 class from compat, none from the reference
    dep = subprocess.run(["g++", "-std=c++17", "-w", "-x", "c++", "-MM", "-I", os.path.join(ROOT, "dftatom_amd", "compat"), "-I", REF, "-"],
                         stdin=open(os.path.join(REF, "DFTAtom.cpp")), capture_output=True, text=True, check=True).stdout
    used = {os.path.basename(p): os.path.dirname(os.path.abspath(p)) for p in dep.replace("\\\n", " ").split() if p.endswith(".h")}
    compat = os.path.join(ROOT, "dftatom_amd", "compat")
    for h in ("Numerov.h", "PoissonSolver.h", "VWNExcCor.h", "Integral.h", "AufbauPrinciple.h", "DFTAtom.h", "ExcCor.h"):
        assert used[h] == compat, (h, used[h])
    assert not any(d == REF for d in used.values()), used
    # the object defines the reference's orchestration and imports the C ABI
    syms = subprocess.run(["nm", "-C", exe], capture_output=True, text=True, check=True).stdout
    for s in ("DFT::DFTAtom::LoopOverLevels(DFT::Numerov<DFT::NumerovFunctionNonUniformGrid>&",
              "DFT::DFTAtom::LoopOverLevels(DFT::Numerov<DFT::NumerovFunctionRegularGrid>&",
              "DFT::DFTAtom::CalculateUniformLSDA(int, int, double, double)", "DFT::DFTAtom::NormalizeNonUniform("):
        assert s in syms, s
    for s in ("U dfta_potential_sweeps", "U dfta_potential_match", "U dfta_potential_update", "U dfta_poisson_solve", "U dfta_vwn_lda", "U dfta_vwn_lsda",
              "U dfta_integrate", "U dfta_grid_create_uniform"):
        assert s in syms, s
    # without a GPU the binary must fail loudly (no CPU fallback anywhere below the reference's L3)
    r = subprocess.run([exe, "2", "10", "0.5", "10", "0.01", "0"], capture_output=True, text=True)
    if r.returncode == 0:
        pytest.skip("a HIP device is present")
    assert "no usable HIP device" in r.stderr
