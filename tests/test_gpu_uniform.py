"""GPU suite (-m gpu): the uniform-grid path -- NumerovFunctionRegularGrid (Numerov.h:16-70 and the
IsUniform() branches of Numerov.h:272-504), SolvePoissonUniform (PoissonSolver.h:20-49), NormalizeUniform and the uniform
LoopOverLevels (DFTAtom.cpp:21-33, 213-325), CalculateUniformLDA / LSDA (DFTAtom.cpp:60-210, 646-844) -- against vectors
captured from the compiled reference (tests/golden/make_golden_table.py uniform).

Tolerances: sweeps / match with host boundary values bit-exact; level driver 1e-10 Ha (device exp() in the start values);
Poisson 1e-10 Z; SCF steps 1e-9 relative (energies), 1e-8 Ha + 1e-10 |E| (eigenvalues).
"""
import json
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import dftatom_amd as D                 # noqa: E402
from golden.make_golden import screened_potential   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def ctx(torch_first):
    c = D.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "uniform.npz")), json.load(open(os.path.join(HERE, "golden", "uniform_meta.json")))


@pytest.fixture(scope="module")
def ugrid(ctx, gold):
    m = gold[1]["grid"]
    g = D.Grid(ctx, m["L"], None, m["Rmax"])
    assert g.uniform and g.N == m["N"]
    yield g
    g.close()


def _pots(g):
    rr = g.r()
    return {"coulomb10": np.concatenate([[0.0], -10.0 / rr[1:]]), "screened18": screened_potential(rr, 18.0)}


def test_uniform_grid_positions(ugrid, gold):
    m = gold[1]["grid"]
    assert np.array_equal(ugrid.r(), m["h"] * np.arange(m["N"]))        # position = h * i (Numerov.h:313)


@pytest.mark.parametrize("pname", ["coulomb10", "screened18"])
def test_uniform_sweeps_bit_exact(ctx, ugrid, gold, pname):
    data = gold[0]
    V = _pots(ugrid)[pname]
    rows = data["sweeps_" + pname]                    # l, E, limit, count, u0
    c = D.numerov_sweeps(ctx, ugrid, D.SWEEP_COUNT, V, rows[:, 0], rows[:, 1], rows[:, 2])
    assert np.array_equal(c["count"], rows[:, 3].astype(np.int32))
    z = D.numerov_sweeps(ctx, ugrid, D.SWEEP_ZERO, V, rows[:, 0], rows[:, 1])
    assert np.array_equal(z["u0"].view(np.int64), rows[:, 4].copy().view(np.int64))
    m = data["match_" + pname]
    psi, mp = D.numerov_match(ctx, ugrid, V, m[:, 0], m[:, 1])
    assert np.array_equal(mp, m[:, 2].astype(np.int64))
    assert np.array_equal(np.nansum(psi, axis=1), m[:, 3]) and np.array_equal(np.nansum(np.abs(psi), axis=1), m[:, 4])
    assert np.array_equal(psi[:, :: ugrid.N // 64][:, :64], m[:, 5:], equal_nan=True)
    # device-side start values (device exp / pow): counts unchanged, u(0) to 1e-10 relative
    cd = D.numerov_sweeps(ctx, ugrid, D.SWEEP_COUNT, V, rows[:, 0], rows[:, 1], rows[:, 2], boundary=D.BOUNDARY_DEVICE)
    assert np.array_equal(cd["count"], c["count"])
    zd = D.numerov_sweeps(ctx, ugrid, D.SWEEP_ZERO, V, rows[:, 0], rows[:, 1], boundary=D.BOUNDARY_DEVICE)
    ok = np.isfinite(rows[:, 4]) & (rows[:, 4] != 0)
    assert np.max(np.abs(zd["u0"][ok] - rows[ok, 4]) / np.abs(rows[ok, 4])) <= 1e-10


def test_uniform_level_driver_vs_golden(ctx, ugrid, gold):
    data = gold[0]
    V = _pots(ugrid)["screened18"]
    lv = D.get_subshells(18)
    res = D.solve_levels(ctx, ugrid, V, lv, -18.0 * 18 - 1.0, mode=D.LEVELS_CHAINED)
    assert np.max(np.abs(res["E"] - data["levels_E"])) <= 1e-10
    assert res["converged"].all() == bool(data["levels_scalars"][2])
    nd = res["newDensity"][0]
    want = data["levels_newdensity_sample"]
    got = nd[:: ugrid.N // 256]
    assert np.max(np.abs(got - want)) <= 1e-9 * np.max(np.abs(want))
    assert abs(res["Eelectronic"][0] - data["levels_scalars"][0]) <= 1e-9 * abs(data["levels_scalars"][0])
    # the un-chained default mode finds the same levels
    res_b = D.solve_levels(ctx, ugrid, V, lv, -18.0 * 18 - 1.0, mode=D.LEVELS_BATCHED)
    assert np.max(np.abs(res_b["E"] - data["levels_E"]) / np.abs(data["levels_E"])) <= 1e-9


@pytest.mark.parametrize("tag,Z", [("Z2", 2), ("Z18", 18)])
def test_uniform_poisson_vs_golden(ctx, gold, tag, Z):
    data, meta = gold
    pg = meta["poisson_grid"]
    g = D.Grid(ctx, pg["L"], None, pg["Rmax"])
    rr = g.r()
    ps = D.Poisson(ctx, g, 1)
    U, vc, err = ps.solve([Z], Z * np.exp(-2 * rr) / np.pi)
    want = data["poisson_%s_U" % tag]
    print("uniform Poisson %s: max |dU| %.2e, V-cycles %d" % (tag, np.max(np.abs(U[0] - want)), vc[0]))
    assert np.max(np.abs(U[0] - want)) <= 1e-10 * Z
    # FullCycle again on the same source with the same boundaries: the same end state (PoissonSolver.h:89-124)
    err2, vc2 = ps.full_cycle(0.0, float(Z))
    phi, _ = ps.get_level(0)
    assert vc2 == vc[0] and np.array_equal(phi.view(np.int64), U[0].view(np.int64))
    ps.close()
    g.close()


@pytest.mark.parametrize("tag,lsda", [("Ne_uLDA_L12", False), ("N_uLSDA_L12", True)])
def test_uniform_scf_vs_reference(ctx, gold, tag, lsda):
    ref = gold[1]["end_to_end"][tag]
    g = D.Grid(ctx, ref["L"], None, ref["Rmax"])
    scf = D.Scf(ctx, g, [ref["Z"]], lsda=lsda, levels_mode=D.LEVELS_CHAINED)
    worst_lv, worst_en = 0.0, 0.0
    for k in range(3):
        scf.step()
        en, fin = scf.energies()
        want_lv = np.array([x[1] for x in ref["steps"][k]["levels"]])
        got_lv = np.concatenate([scf.levels(0, 0)["E"]] + ([scf.levels(0, 1)["E"]] if lsda else []))
        dlv = np.abs(got_lv - want_lv)
        den = np.array([abs(a - b) / abs(b) for a, b in zip(en[0].as_list(), ref["steps"][k]["energies"])])
        worst_lv, worst_en = max(worst_lv, dlv.max()), max(worst_en, den.max())
        assert np.all(dlv <= 1e-8 + 1e-10 * np.abs(want_lv)), (k, dlv)
        assert np.all(den <= 1e-9), (k, den)
    scf.close()
    print("%s: first steps max |dE_level| %.2e Ha, energies %.2e rel" % (tag, worst_lv, worst_en))
    # to the reference's stop, default (un-chained) mode: every step's Etotal, then the stop state
    want = np.array(ref["etotal_all"])
    scf = D.Scf(ctx, g, [ref["Z"]], lsda=lsda)
    traj = []
    for _ in range(len(want)):
        scf.step(want_stats=False)
        traj.append(scf.energies()[0][0].Etotal)
        if scf.energies()[1][0]:
            break
    traj = np.array(traj)
    n = min(len(traj), len(want))
    rel = np.abs(traj[:n] - want[:n]) / np.abs(want[:n])
    print("%s: %d steps (reference %d), trajectory max %.2e rel" % (tag, len(traj), len(want), rel.max()))
    assert rel.max() <= 1e-9
    last = ref["steps"][-1]
    for a, b in zip(scf.energies()[0][0].as_list(), last["energies"]):
        assert abs(a - b) <= 2e-9 * abs(b)
    scf.close()
    g.close()


def test_headless_front_end_uniform_text(gold):
    """CalculateUniformLDA through dftatom_cli (method 2): the reference's banner, tagged level lines, final configuration"""
    ref = gold[1]["end_to_end"]["Ne_uLDA_L12"]
    exe = os.path.join(ROOT, "dftatom_amd", "compat", "dftatom_cli")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    out = subprocess.run([exe, "10", "12", "0.5", "15", "0", "2"], check=True, capture_output=True, text=True, timeout=600).stdout
    lines = out.strip().splitlines()
    assert lines[0] == ref["banner"] == "Computing atom with Z=10 using LDA with uniform grid"
    assert "Finished!" in out and lines[-1].strip() == "1s2 2s2 2p6"
    got = [float(re.search(r": (\S+) Num", ln).group(1)) for ln in lines if ln.startswith("Energy")][-3:]
    want = [round(x[1], 6) for x in ref["steps"][-1]["levels"]]
    assert np.max(np.abs(np.array(got) - np.array(want))) <= 1.5e-6


# ---------------------------------------------------------------------------------------------------------------
# the reference's compile-time alternatives as run-time options
# ---------------------------------------------------------------------------------------------------------------
def test_chachiyo_vs_reference(ctx, gold):
    """ChachiyoExchCor<Param>::Vexc / eexcDif (ExcCor.h:27-95), both parameter sets: 1e-12 relative (device pow / log)"""
    data = gold[0]
    n = data["chachiyo_n"]
    for imp in (0, 1):
        v, e = D.chachiyo_lda(ctx, n, improved=bool(imp))
        want = data["chachiyo_%d" % imp]
        for got, w in ((v, want[0]), (e, want[1])):
            nz = w != 0
            assert np.array_equal(got[~nz], w[~nz])                      # below 1e-18 the reference returns 0
            assert np.max(np.abs(got[nz] - w[nz]) / np.abs(w[nz])) <= 1e-12
    # a live SCF with it: Ne on the logarithmic grid, total energy within 0.1% of the VWN one (a different fit of the same gas)
    g = D.Grid(ctx, 12, 2e-3, 25.0)
    ea = {}
    for fx in (D.XC_VWN, D.XC_CHACHIYO_IMPROVED):
        scf = D.Scf(ctx, g, [10], functional=fx)
        for _ in range(40):
            scf.step(want_stats=False)
            if scf.energies()[1][0]:
                break
        ea[fx] = scf.energies()[0][0].Etotal
        scf.close()
    assert abs(ea[D.XC_CHACHIYO_IMPROVED] - ea[D.XC_VWN]) < 1e-3 * abs(ea[D.XC_VWN]) and ea[D.XC_CHACHIYO_IMPROVED] != ea[D.XC_VWN]
    with pytest.raises(D.DftaError):
        D.Scf(ctx, g, [10], lsda=True, functional=D.XC_CHACHIYO)         # LDA only, as in the reference
    g.close()


def test_live_integrator_switch(ctx):
    """Romberg / Boole / Simpson 1/3 / trapezoid as the live quadrature of an SCF (energy integrals + normalisation):
    every rule converges Ne to the same energies to quadrature accuracy; Simpson38 is the default and bit-identical to it."""
    g = D.Grid(ctx, 12, 2e-3, 25.0)
    res = {}
    for rule in (None, D.INT_SIMPSON38, D.INT_ROMBERG, D.INT_BOOLE, D.INT_SIMPSON13, D.INT_TRAPEZOID):
        scf = D.Scf(ctx, g, [10]) if rule is None else D.Scf(ctx, g, [10], integrator=rule)
        for _ in range(3):
            scf.step(want_stats=False)
        res[rule] = scf.energies()[0][0].as_list()
        scf.close()
    assert res[None] == res[D.INT_SIMPSON38]
    for rule in (D.INT_ROMBERG, D.INT_BOOLE, D.INT_SIMPSON13, D.INT_TRAPEZOID):
        assert res[rule] != res[D.INT_SIMPSON38]
        assert np.allclose(res[rule], res[D.INT_SIMPSON38], rtol=2e-5, atol=0), rule     # quadrature error of the rules on 4097 nodes
    # set_integrator on a running SCF == created with it (the rule does not enter the start potential)
    a = D.Scf(ctx, g, [10])
    a.set_integrator(D.INT_ROMBERG)
    for _ in range(3):
        a.step(want_stats=False)
    assert a.energies()[0][0].as_list() == res[D.INT_ROMBERG]
    a.close()
    g.close()


def test_transition_metal_option_vs_oracle(ctx):
    """Cr with AdjustForTransitionMetals wired in (3d5 4s1, AufbauPrinciple.h:78-99) vs the reference's plain Madelung filling
    (3d4 4s2): five SCF steps of BOTH configurations against the oracle run on the same level list (chained brackets, the
    reference's path) -- energies 1e-9 relative, eigenvalues 1e-8 Ha + 2e-9 |E| (the test only used to run)."""
    import ctypes as C
    import _oracle as O
    L, d, R = 12, 2e-3, 25.0
    g = D.Grid(ctx, L, d, R)
    ref_cfg = D.get_subshells(24)
    tm_cfg = D.get_subshells(24, D.AUFBAU_TRANSITION_METALS)
    assert (2, 2, 4) in ref_cfg and (3, 0, 2) in ref_cfg and (2, 2, 5) in tm_cfg and (3, 0, 1) in tm_cfg
    o = O.oracle()
    e = {}
    for au, cfg in ((D.AUFBAU_REFERENCE, ref_cfg), (D.AUFBAU_TRANSITION_METALS, tm_cfg)):
        scf = D.Scf(ctx, g, [24], aufbau=au, levels_mode=D.LEVELS_CHAINED)
        s = o.dfo_scf_create(0, 24, L, 0.5, R, d, 1)
        assert s.contents.nla == len(cfg)
        for i, (n, l, occ) in enumerate(cfg):                  # the oracle's level list := the configuration under test
            s.contents.la[i].n, s.contents.la[i].l, s.contents.la[i].occ = n, l, occ
        oe = O.Energies()
        for step in range(5):
            scf.step(want_stats=False)
            o.dfo_scf_step(s, C.byref(oe))
            got = scf.energies()[0][0].as_list()
            want = [oe.Etotal, oe.Ekinetic, oe.Ecoul, oe.Enuclear, oe.Exc]
            for a, b in zip(got, want):
                assert abs(a - b) <= 1e-9 * abs(b), (au, step, got, want)
            lv = scf.levels(0, 0)
            lo = np.array([s.contents.la[i].E for i in range(s.contents.nla)])
            assert np.all(np.abs(lv["E"] - lo) <= 1e-8 + 2e-9 * np.abs(lo)), (au, step)
        assert sum(lv["occ"]) == 24
        e[au] = scf.energies()[0][0].Etotal
        o.dfo_scf_destroy(s)
        scf.close()
    assert e[0] != e[1] and abs(e[0] - e[1]) < 1.0
    g.close()
