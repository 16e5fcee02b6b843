"""CPU suite: the C-ABI shared library loads and exports every symbol include/dftatom_hip.h declares; the
integer-only host entry points work; there is NO CPU fallback (context creation fails without a HIP device)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import dftatom_amd as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "dftatom_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dfta_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    names = _declared()
    assert len(names) >= 40
    lib = D.load()                        # binds every entry of SIGNATURES, raises on a missing symbol
    for n in names:
        assert hasattr(lib, n), "declared in the header but not exported: " + n
        assert n in D.SIGNATURES, "declared in the header but not bound in dftatom_amd.SIGNATURES: " + n
    assert sorted(D.SIGNATURES) == names


def test_struct_layouts_match_the_header():
    """the four plain structs that cross the boundary: the ctypes mirrors have the header's fields, in its order, with its types"""
    src = open(os.path.join(ROOT, "include", "dftatom_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    ctype = {"long": C.c_long, "int": C.c_int, "float": C.c_float, "double": C.c_double}
    mirrors = {"dfta_level_result": D.LevelResult, "dfta_energies": D.Energies, "dfta_step_stats": D.StepStats,
               "dfta_scf_options": D.ScfOptions}
    found = {}
    for body, name in re.findall(r"typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            t, rest = decl.split(None, 1)
            fields += [(n.strip(), ctype[t]) for n in rest.split(",")]
        found[name] = fields
    assert sorted(found) == sorted(mirrors)
    for name, cls in mirrors.items():
        assert [(n, t) for n, t in cls._fields_] == found[name], name


def test_signatures_cite_reference():
    """every block of the header names the reference interface it replaces (file:line)"""
    src = open(os.path.join(ROOT, "include", "dftatom_hip.h")).read()
    for token in ("Numerov.h:272-349", "Numerov.h:351-401", "Numerov.h:403-504", "DFTAtom.cpp:36-56", "PoissonSolver.h:51-81",
                  "PoissonSolver.cpp:40-64", "VWNExcCor.h:73-128", "VWNExcCor.h:134-312", "Integral.h:11-155",
                  "DFTAtom.cpp:346-491", "AufbauPrinciple.h:36-75"):
        assert token in src, token


def test_host_only_entry_points(golden):
    _, meta = golden
    lib = D.load()
    assert [lib.dfta_num_nodes(L) for L in (3, 14, 17, 20)] == [9, 16385, 131073, 1048577]
    for Z in range(1, 119):
        assert [list(t) for t in D.get_subshells(Z)] == meta["aufbau"][str(Z)], Z
    # LSDA occupation split (DFTAtom.cpp:611-638): nitrogen 1s 2s 2p3 -> alpha (1,1,3), beta (1,1)
    arrs = [np.zeros(32, np.int32) for _ in range(6)]
    nA, nB = C.c_int(), C.c_int()
    assert lib.dfta_split_spin(7, C.byref(nA), C.byref(nB), *[a.ctypes.data_as(D.c_ip) for a in arrs], 32) == D.OK
    assert (nA.value, nB.value) == (3, 2)
    assert arrs[2][:3].tolist() == [1, 1, 3] and arrs[5][:2].tolist() == [1, 1]


def test_no_cpu_fallback():
    """Without a HIP device the product path refuses to run (it never routes through the oracle)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = D.load()
    h = C.c_void_p()
    assert lib.dfta_ctx_create(0, None, C.byref(h)) == 2       # DFTA_ERR_NO_DEVICE
    with pytest.raises(D.DftaError):
        D.Context(0)
    # the product package has no reference to the oracle
    for root, _, files in os.walk(os.path.join(ROOT, "dftatom_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "dfta_oracle" not in txt and "libdfta_ref" not in txt and "_oracle" not in txt, f


def test_aufbau_transition_metal_option():
    """AufbauPrinciple::AdjustForTransitionMetals (AufbauPrinciple.h:78-99, unwired in the reference) as an option: the known
    ground-state configurations of the d-block exceptions; every other atom is untouched.  Integer-only host code."""
    import dftatom_amd as D
    name = "spdf"

    def cfg(Z, au):
        return " ".join("%d%s%d" % (n + 1, name[l], occ) for n, l, occ in D.get_subshells(Z, au))
    want = {24: ("3d4 4s2", "3d5 4s1"), 29: ("3d9 4s2", "3d10 4s1"), 41: ("4d3 5s2", "4d4 5s1"), 42: ("4d4 5s2", "4d5 5s1"),
            44: ("4d6 5s2", "4d7 5s1"), 45: ("4d7 5s2", "4d8 5s1"), 46: ("4d8 5s2", "4d10"), 47: ("4d9 5s2", "4d10 5s1"),
            78: ("5d8 6s2", "5d9 6s1"), 79: ("5d9 6s2", "5d10 6s1")}
    for Z in range(1, 119):
        ref, tm = cfg(Z, D.AUFBAU_REFERENCE), cfg(Z, D.AUFBAU_TRANSITION_METALS)
        assert sum(o for _, _, o in D.get_subshells(Z, D.AUFBAU_TRANSITION_METALS)) == Z
        if Z in want:
            for part in want[Z][0].split():
                assert part in ref.split(), (Z, ref)
            for part in want[Z][1].split():
                assert part in tm.split(), (Z, tm)
            if Z == 46:
                assert "5s" not in tm
        else:
            assert ref == tm, Z


def test_transition_metal_aufbau_option():
    """dfta_get_subshells_ex(..., DFTA_AUFBAU_TRANSITION_METALS): the reference's unwired AdjustForTransitionMetals
    (AufbauPrinciple.h:78-99: the d shell takes one electron from the s shell above it for Cr, Cu, Nb, Mo, Ru, Rh, Ag, Pt, Au; two for
    Pd) as an option -- the textbook ground-state configurations, every electron accounted for, and no change for the other atoms
    (host code: no device needed)."""
    import dftatom_amd as D
    lab = "spdf"

    def cfg(Z, a):
        return " ".join("%d%s%d" % (n + 1, lab[l], o) for n, l, o in D.get_subshells(Z, a))

    want = {24: "3d5 4s1", 29: "3d10 4s1", 41: "4d4 5s1", 42: "4d5 5s1", 44: "4d7 5s1", 45: "4d8 5s1", 46: "4p6 4d10", 47: "4d10 5s1",
            78: "5d9 6s1", 79: "5d10 6s1"}
    for Z in range(1, 119):
        ref, tm = cfg(Z, D.AUFBAU_REFERENCE), cfg(Z, D.AUFBAU_TRANSITION_METALS)
        assert sum(o for _, _, o in D.get_subshells(Z, D.AUFBAU_TRANSITION_METALS)) == Z
        if Z in want:
            assert tm.endswith(want[Z]) and tm != ref, (Z, tm)
        else:
            assert tm == ref, (Z, tm, ref)
