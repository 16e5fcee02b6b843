"""CPU suite: the oracle against the COMPILED REFERENCE itself (oracle/_ref/libdfta_ref.so, built from
/root/reference by `make -C oracle ref`).  Skipped where the reference build is absent.
Everything here is bit-exact: the oracle is a restatement, not an approximation."""
import ctypes as C

import numpy as np
import pytest

import _oracle as O
from golden.make_golden import parse_run, ref_text, screened_potential

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (no /root/reference here)")


def test_numerov_random_trials():
    o, r = O.oracle(), O.ref()
    g = O.make_grid(13, 1e-3, 30.0)
    rr = O.grid_r(g)
    rng = np.random.default_rng(7)
    for Z, V in ((18, O.coulomb_potential(g, 18)), (50, screened_potential(rr, 50.0))):
        h = r.ref_numerov_create(O.dp(V), g.N, g.delta, g.Rmax)
        assert r.ref_rp(h) == g.Rp
        Es = np.concatenate([-rng.uniform(1e-4, Z * Z + 1, 40), [0.3, 49.0, -1e-6]])
        P1, P2 = np.zeros(g.N), np.zeros(g.N)
        for l in range(4):
            for E in Es:
                E = float(E)
                assert o.dfo_max_radius_index(C.byref(g), E, g.N - 1) == r.ref_max_radius_index(h, E)
                for lim in (0, 1, 4):
                    assert o.dfo_count_nodes(C.byref(g), O.dp(V), l, E, lim, None, None) == r.ref_count_nodes(h, l, E, lim)
                a, b = o.dfo_solution_in_zero(C.byref(g), O.dp(V), l, E, None), r.ref_solution_in_zero(h, l, E)
                assert a == b or (np.isnan(a) and np.isnan(b))
                assert o.dfo_match(C.byref(g), O.dp(V), l, E, O.dp(P1), None) == r.ref_match(h, l, E, O.dp(P2))
                assert np.array_equal(P1, P2, equal_nan=True)
        r.ref_numerov_destroy(h)


def test_level_driver_and_normalise():
    o, r = O.oracle(), O.ref()
    g = O.make_grid(12, 2e-3, 25.0)
    rr = O.grid_r(g)
    V = screened_potential(rr, 36.0)
    lv = O.subshells(36)
    h = r.ref_numerov_create(O.dp(V), g.N, g.delta, g.Rmax)
    n = np.array([a for a, _, _ in lv], np.int32)
    l = np.array([b for _, b, _ in lv], np.int32)
    occ = np.array([c for _, _, c in lv], np.int32)
    E2, nd2, Eel2, Bot2 = np.zeros(len(lv)), np.zeros(g.N), C.c_double(0), C.c_double(-36.0 * 36 - 1)
    c2 = r.ref_loop_over_levels(h, len(lv), O.ip(n), O.ip(l), O.ip(occ), O.dp(E2), O.dp(nd2), C.byref(Eel2), C.byref(Bot2), g.delta)
    arr, nd1, Eel1, Bot1 = O.levels_array(lv), np.zeros(g.N), C.c_double(0), C.c_double(-36.0 * 36 - 1)
    c1 = o.dfo_loop_over_levels(C.byref(g), O.dp(V), arr, len(lv), O.dp(nd1), C.byref(Eel1), C.byref(Bot1), 1, None)
    assert np.array_equal(np.array([arr[i].E for i in range(len(lv))]), E2)
    assert np.array_equal(nd1, nd2) and Eel1.value == Eel2.value and Bot1.value == Bot2.value and c1 == c2
    psi = np.random.default_rng(3).standard_normal(g.N)
    a, b = psi.copy(), psi.copy()
    o.dfo_normalize_nonuniform(C.byref(g), O.dp(a))
    r.ref_normalize_nonuniform(O.dp(b), g.N, g.Rp, g.delta)
    assert np.array_equal(a, b)
    r.ref_numerov_destroy(h)


def test_poisson_full_solves():
    o, r = O.oracle(), O.ref()
    for L, d, R, Z in ((11, 4e-3, 25.0, 1), (13, 1e-3, 25.0, 18), (14, 5e-4, 25.0, 86)):
        g = O.make_grid(L, d, R)
        rr = O.grid_r(g)
        rho = Z * np.exp(-2 * rr) / np.pi
        p, q = o.dfo_poisson_create(L, d), r.ref_poisson_create(L, d)
        U1, U2 = np.zeros(g.N), np.zeros(g.N)
        for _ in range(2):   # second call exercises state carried in the solver object
            o.dfo_solve_poisson_nonuniform(p, Z, R, O.dp(rho), O.dp(U1))
            r.ref_solve_poisson_nonuniform(q, Z, R, O.dp(rho), g.N, O.dp(U2))
            assert np.array_equal(U1, U2)
            rho = rho * 0.9 + 0.01 * np.exp(-rr)
        o.dfo_poisson_destroy(p)
        r.ref_poisson_destroy(q)


@pytest.mark.parametrize("mode", [0, 1])
def test_scf_small_atom_every_step(mode):
    """Every SCF step of a 12-level Argon run: all eigenvalues and energy terms identical to the
    reference's 17-digit console output, same number of steps, same Finished! decision."""
    o, r = O.oracle(), O.ref()
    txt = ref_text(r, mode, 18, 12, 0.5, 25.0, 2e-3, hp=True)
    steps = parse_run(txt)
    s = o.dfo_scf_create(mode, 18, 12, 0.5, 25.0, 2e-3, 1)
    e = O.Energies()
    fin = 0
    for k, st in enumerate(steps):
        assert not fin
        fin = o.dfo_scf_step(s, C.byref(e))
        lv = [s.contents.la[i].E for i in range(s.contents.nla)]
        if mode:
            lv += [s.contents.lb[i].E for i in range(s.contents.nlb)]
        assert lv == [x[1] for x in st["levels"]], k
        assert [e.Etotal, e.Ekinetic, e.Ecoul, e.Enuclear, e.Exc] == st["energies"], k
    assert bool(fin) == ("Finished!" in txt)
    o.dfo_scf_destroy(s)


def test_open_shell_lsda_differs_from_lda_and_matches_ref():
    """Z=7 (open p shell): LSDA splits alpha/beta eigenvalues; three steps, exact vs reference."""
    o, r = O.oracle(), O.ref()
    txt = ref_text(r, 1, 7, 11, 0.5, 25.0, 4e-3, hp=True)
    steps = parse_run(txt)
    s = o.dfo_scf_create(1, 7, 11, 0.5, 25.0, 4e-3, 1)
    e = O.Energies()
    for k in range(3):
        o.dfo_scf_step(s, C.byref(e))
        lv = [s.contents.la[i].E for i in range(s.contents.nla)] + [s.contents.lb[i].E for i in range(s.contents.nlb)]
        assert lv == [x[1] for x in steps[k]["levels"]]
        assert [e.Etotal, e.Ekinetic, e.Ecoul, e.Enuclear, e.Exc] == steps[k]["energies"]
    assert s.contents.nla == 3 and s.contents.nlb == 2
    o.dfo_scf_destroy(s)


def test_uniform_grid_restatement_bit_exact_vs_reference():
    """The oracle's uniform-grid functions (NumerovFunctionRegularGrid sweeps and match incl. the re-derived step, the
    uniform LoopOverLevels / NormalizeUniform, SolvePoissonUniform) against the compiled reference: bit for bit."""
    r, o = O.ref(), O.oracle()
    L, R = 12, 25.0
    g = O.make_ugrid(L, R)
    N = g.N
    rr = g.h * np.arange(N)
    rng = np.random.default_rng(11)
    for Z, V in ((10.0, np.concatenate([[0.0], -10.0 / rr[1:]])), (18.0, screened_potential(rr, 18.0))):
        hd = r.ref_unumerov_create(O.dp(V), N, R)
        Es = np.concatenate([-rng.uniform(1e-3, Z * Z + 1, 12), [-(Z ** 2) / 2, -33.0, -31.0, -1e-3, 0.5, 50.0]])
        Pr, Po = np.zeros(N), np.zeros(N)
        for l in range(4):
            for E in Es:
                for lim in (0, 3):
                    assert o.dfo_ucount_nodes(C.byref(g), O.dp(V), l, float(E), lim, None) == r.ref_ucount_nodes(hd, l, float(E), lim)
                a, b = o.dfo_usolution_in_zero(C.byref(g), O.dp(V), l, float(E)), r.ref_usolution_in_zero(hd, l, float(E))
                assert a == b or (np.isnan(a) and np.isnan(b))
                assert o.dfo_umatch(C.byref(g), O.dp(V), l, float(E), O.dp(Po)) == r.ref_umatch(hd, l, float(E), O.dp(Pr))
                assert np.array_equal(Po, Pr, equal_nan=True)
        # level driver
        lv = O.subshells(int(Z))
        n = np.array([a for a, _, _ in lv], np.int32)
        l_ = np.array([b for _, b, _ in lv], np.int32)
        occ = np.array([c for _, _, c in lv], np.int32)
        Er, ndr = np.zeros(len(lv)), np.zeros(N)
        eel_r, bot_r = C.c_double(0), C.c_double(-Z * Z - 1.0)
        conv_r = r.ref_uloop_over_levels(hd, len(lv), O.ip(n), O.ip(l_), O.ip(occ), O.dp(Er), O.dp(ndr), C.byref(eel_r), C.byref(bot_r))
        lev = O.levels_array(lv)
        ndo = np.zeros(N)
        eel_o, bot_o = C.c_double(0), C.c_double(-Z * Z - 1.0)
        conv_o = o.dfo_uloop_over_levels(C.byref(g), O.dp(V), lev, len(lv), O.dp(ndo), C.byref(eel_o), C.byref(bot_o))
        assert conv_o == conv_r and eel_o.value == eel_r.value and bot_o.value == bot_r.value
        assert [lev[k].E for k in range(len(lv))] == list(Er)
        assert np.array_equal(ndo, ndr)
        r.ref_unumerov_destroy(hd)
    # NormalizeUniform alone, and SolvePoissonUniform (deltaGrid = 0)
    psi = rng.standard_normal(N)
    a, b = psi.copy(), psi.copy()
    o.dfo_normalize_uniform(O.dp(a), N, g.h)
    r.ref_normalize_uniform(O.dp(b), N, g.h)
    assert np.array_equal(a, b)
    for Z in (2, 18):
        rho = Z * np.exp(-2 * rr) / np.pi
        q = r.ref_poisson_create(L, 0.0)
        p = o.dfo_poisson_create(L, 0.0)
        Ur, Uo = np.zeros(N), np.zeros(N)
        r.ref_solve_poisson_uniform(q, Z, R, O.dp(rho), N, O.dp(Ur))
        o.dfo_solve_poisson_uniform(p, Z, R, O.dp(rho), O.dp(Uo))
        assert np.array_equal(Uo, Ur)
        r.ref_poisson_destroy(q)
        o.dfo_poisson_destroy(p)
