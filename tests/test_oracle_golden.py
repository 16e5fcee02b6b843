"""CPU suite: the oracle (oracle/dfta_oracle.c) against golden vectors captured from the compiled
reference (tests/golden/make_golden.py).  Bit-exact unless a tolerance is written in the test."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import _oracle as O
from golden.make_golden import GRIDS, screened_potential


def test_grid_f_tables(golden):
    data, meta = golden
    o = O.oracle()
    for name, (L, d, R) in GRIDS.items():
        g = O.make_grid(L, d, R)
        m = meta[f"grid_{name}"]
        assert g.N == m["N"] and g.Rp == m["Rp"]
        V0 = np.zeros(g.N)
        idx = data[f"grid_{name}_idx"]
        f0 = np.array([o.dfo_f(C.byref(g), O.dp(V0), 0, -1.0, int(i)) for i in idx])
        # index 0 has r=0: l=0 centrifugal term is 0/0 -> NaN in the reference too
        assert np.array_equal(f0, data[f"grid_{name}_f_l0_Em1"], equal_nan=True)
        f3 = np.array([o.dfo_f(C.byref(g), O.dp(V0), 3, -2.5, int(i)) for i in idx[1:]])
        assert np.array_equal(f3, data[f"grid_{name}_f_l3_Em2p5"])


def _pots():
    L, d, R = GRIDS["L14"]
    g = O.make_grid(L, d, R)
    rr = O.grid_r(g)
    return g, {"coulomb18": O.coulomb_potential(g, 18), "screened18": screened_potential(rr, 18.0),
               "screened86": screened_potential(rr, 86.0)}


@pytest.mark.parametrize("pname", ["coulomb18", "screened18", "screened86"])
def test_numerov_sweeps_bit_exact(golden, pname):
    data, _ = golden
    o = O.oracle()
    g, pots = _pots()
    V = pots[pname]
    if pname == "screened18":
        assert np.array_equal(V, data["pot_screened18_L14"])   # numpy/libm determinism of the input itself
    for l, E, lim, cnt in data[f"numerov_{pname}_counts"]:
        got = o.dfo_count_nodes(C.byref(g), O.dp(V), int(l), float(E), int(lim), None, None)
        assert got == int(cnt), (pname, l, E, lim)
    P = np.zeros(g.N)
    for row in data[f"numerov_{pname}_sweeps"]:
        l, E, u0, cut, mp, s1, s2 = row[:7]
        st = C.c_long()
        got = o.dfo_solution_in_zero(C.byref(g), O.dp(V), int(l), float(E), C.byref(st))
        assert got == u0 or (np.isnan(got) and np.isnan(u0))
        assert st.value == int(cut)
        m = o.dfo_match(C.byref(g), O.dp(V), int(l), float(E), O.dp(P), None)
        assert m == int(mp)
        assert np.nansum(P) == s1 and np.nansum(np.abs(P)) == s2
        assert np.array_equal(P[:: max(1, g.N // 64)], row[7:], equal_nan=True)


@pytest.mark.parametrize("pname,Z", [("screened18", 18), ("screened86", 86)])
def test_level_driver_bit_exact(golden, pname, Z):
    data, _ = golden
    o = O.oracle()
    g, pots = _pots()
    V = pots[pname]
    lv = O.subshells(Z)
    arr = O.levels_array(lv)
    nd = np.zeros(g.N)
    Eel = C.c_double(0)
    Bot = C.c_double(-float(Z) * Z - 1.0)
    conv = o.dfo_loop_over_levels(C.byref(g), O.dp(V), arr, len(lv), O.dp(nd), C.byref(Eel), C.byref(Bot), 1, None)
    E = np.array([arr[i].E for i in range(len(lv))])
    assert np.array_equal(E, data[f"levels_{pname}_E"])
    assert np.array_equal(nd[:: g.N // 256], data[f"levels_{pname}_newdensity_sample"])
    assert np.array_equal(np.array([Eel.value, Bot.value, conv, nd.sum()]), data[f"levels_{pname}_scalars"])
    for n, l, top, bot in data[f"levels_{pname}_locate"]:
        t = C.c_double(50.0)
        b = C.c_double(-float(Z) * Z - 1.0)
        o.dfo_locate_interval(C.byref(g), O.dp(V), C.byref(t), C.byref(b), int(l), int(n - l), 1e-12, None)
        assert (t.value, b.value) == (top, bot)


def test_level_driver_unchained_is_close(golden):
    """The batched (un-chained) bracket start is a documented deviation: every level starts from
    -Z^2-1 instead of E_prev-3 (DFTAtom.cpp:541).  Eigenvalues move by less than the bracket
    tolerance class: |dE| <= 1e-10 (stated tolerance)."""
    data, _ = golden
    o = O.oracle()
    g, pots = _pots()
    lv = O.subshells(18)
    arr = O.levels_array(lv)
    nd = np.zeros(g.N)
    Eel = C.c_double(0)
    Bot = C.c_double(-18.0 * 18 - 1.0)
    o.dfo_loop_over_levels(C.byref(g), O.dp(pots["screened18"]), arr, len(lv), O.dp(nd), C.byref(Eel), C.byref(Bot), 0, None)
    E = np.array([arr[i].E for i in range(len(lv))])
    assert np.max(np.abs(E - data["levels_screened18_E"])) <= 1e-10


@pytest.mark.parametrize("tag,Z", [("H", 1), ("Z18", 18), ("Z86", 86)])
def test_poisson_solve_bit_exact(golden, tag, Z):
    data, meta = golden
    o = O.oracle()
    m = meta["poisson_grid"]
    g = O.make_grid(m["L"], m["delta"], m["Rmax"])
    rr = O.grid_r(g)
    rho = Z * np.exp(-2 * rr) / np.pi
    p = o.dfo_poisson_create(m["L"], m["delta"])
    U = np.zeros(g.N)
    o.dfo_solve_poisson_nonuniform(p, Z, m["Rmax"], O.dp(rho), O.dp(U))
    o.dfo_poisson_destroy(p)
    assert np.array_equal(U, data[f"poisson_{tag}_U"])
    # known answer: U = Z (1 - (1+r) exp(-2r)) for the 1s density; discretisation error only
    assert np.max(np.abs(U - Z * (1 - (1 + rr) * np.exp(-2 * rr)))) < 3e-7 * Z


def test_multigrid_pieces_bit_exact(golden):
    data, meta = golden
    o = O.oracle()
    Ls, ds = meta["mg_small"]["L"], meta["mg_small"]["delta"]
    p = o.dfo_poisson_create(Ls, ds)

    def view(lvl):
        nl = p.contents.n[lvl]
        return np.ctypeslib.as_array(p.contents.Phi[lvl], (nl,)), np.ctypeslib.as_array(p.contents.Src[lvl], (nl,))

    def check(tag):
        for lvl in range(Ls):
            phi, src = view(lvl)
            assert np.array_equal(phi, data[f"mg_{tag}_phi_{lvl}"]), (tag, lvl)
            assert np.array_equal(src, data[f"mg_{tag}_src_{lvl}"]), (tag, lvl)

    for lvl in range(Ls):
        phi, src = view(lvl)
        phi[:] = data[f"mg_in_phi_{lvl}"]
        src[:] = data[f"mg_in_src_{lvl}"]
    errs = np.array([o.dfo_gauss_seidel(p, lvl) for lvl in range(Ls)])
    assert np.array_equal(errs, data["mg_gs_err"])
    check("gs")
    for lvl in range(1, Ls):
        o.dfo_restrict(p, lvl)
    check("restrict")
    for lvl in range(Ls - 1, 0, -1):
        o.dfo_prolong(p.contents.Phi[lvl], p.contents.n[lvl], p.contents.Phi[lvl - 1])
    check("prolong")
    assert o.dfo_vcycle(p, 1e-14, 3) == data["mg_vcycle_err"][0]
    check("vcycle")
    o.dfo_poisson_destroy(p)


def test_vwn_bit_exact(golden):
    data, _ = golden
    o = O.oracle()
    n = data["vwn_n"]
    a = np.zeros_like(n)
    o.dfo_vwn_vexc(O.dp(n), O.dp(a), len(n))
    assert np.array_equal(a, data["vwn_vexc"])
    o.dfo_vwn_eexcdif(O.dp(n), O.dp(a), len(n))
    assert np.array_equal(a, data["vwn_eexcdif"])
    for zeta in (0.0, 0.3, -0.3, 1.0, -1.0, 0.77):
        na, nb, res, va, vb, e = data[f"vwn_lsda_z{zeta}"]
        r1, a1, b1, e1 = (np.zeros_like(n) for _ in range(4))
        o.dfo_vwn_vexc_lsda(O.dp(np.ascontiguousarray(na)), O.dp(np.ascontiguousarray(nb)), O.dp(r1), O.dp(a1), O.dp(b1), len(n))
        o.dfo_vwn_eexcdif_lsda(O.dp(np.ascontiguousarray(na)), O.dp(np.ascontiguousarray(nb)), O.dp(e1), len(n))
        for x, y in ((r1, res), (a1, va), (b1, vb), (e1, e)):
            assert np.array_equal(x, y, equal_nan=True), zeta
    # LSDA(zeta=0) == LDA up to cancellation in the interpolation terms (measured on this ladder: 1.4e-12)
    na, nb, res, va, vb, e = data["vwn_lsda_z0.0"]
    mask = n >= 1e-18
    assert np.max(np.abs(res[mask] - data["vwn_vexc"][mask]) / np.abs(data["vwn_vexc"][mask])) < 1e-11


def test_integrals_bit_exact(golden):
    data, _ = golden
    o = O.oracle()
    for vec, vals, delta in ((data["int_integrand"], data["int_values"], 1.0), (data["int_noise"], data["int_noise_values"], 0.37)):
        v = np.ascontiguousarray(vec)
        got = [o.dfo_trapezoid(delta, O.dp(v), len(v)), o.dfo_simpson13(delta, O.dp(v), len(v)),
               o.dfo_simpson38(delta, O.dp(v), len(v)), o.dfo_boole(delta, O.dp(v), len(v)),
               o.dfo_romberg(delta, O.dp(v), len(v), 1e-18, 3)]
        assert np.array_equal(np.array(got), vals)
    # the 1s density integrates to one electron (coarse 1025-point grid -> 1e-6 class accuracy)
    assert abs(data["int_values"][2] - 1.0) < 1e-5


def test_aufbau_exact(golden):
    _, meta = golden
    total = 0
    for Z in range(1, 119):
        got = [list(t) for t in O.subshells(Z)]
        assert got == meta["aufbau"][str(Z)], Z
        if Z <= 86:
            total += len(got)
    assert len(O.subshells(86)) == 15
    assert total == 814      # sum of subshells Z=1..86
    # Madelung without the transition-metal exception (AufbauPrinciple.h:78-99 is never called): Cu = 3d9 4s2
    assert (2, 2, 9) in O.subshells(29) and (3, 0, 2) in O.subshells(29)


def _run_scf(mode, Z, L, alpha, R, d, max_steps, chained=1):
    o = O.oracle()
    s = o.dfo_scf_create(mode, Z, L, alpha, R, d, chained)
    e = O.Energies()
    hist = []
    for _ in range(max_steps):
        fin = o.dfo_scf_step(s, C.byref(e))
        lv = [s.contents.la[i].E for i in range(s.contents.nla)]
        if mode:
            lv += [s.contents.lb[i].E for i in range(s.contents.nlb)]
        hist.append((lv, [e.Etotal, e.Ekinetic, e.Ecoul, e.Enuclear, e.Exc]))
        if fin:
            break
    o.dfo_scf_destroy(s)
    return hist


@pytest.mark.parametrize("mode,tag", [(0, "Ar_LDA_L14"), (1, "Ar_LSDA_L14")])
def test_scf_first_steps_match_reference(golden, mode, tag):
    """First three SCF steps of the README Argon configuration (README.md:76), every eigenvalue and energy
    term equal to the reference's 17-digit console output (exact after text round trip)."""
    _, meta = golden
    ref = meta["end_to_end"][tag]
    hist = _run_scf(mode, 18, 14, 0.5, 25.0, 5e-4, 3)
    for k in range(3):
        lv, en = hist[k]
        assert lv == [x[1] for x in ref["steps"][k]["levels"]]
        assert en == ref["steps"][k]["energies"]


@pytest.mark.slow
def test_scf_argon_to_convergence_matches_reference_and_readme(golden):
    """Config 1 of BASELINE.json: Ar LDA, 14 levels, delta 5e-4, Rmax 25, to convergence on the CPU."""
    _, meta = golden
    ref = meta["end_to_end"]["Ar_LDA_L14"]
    hist = _run_scf(0, 18, 14, 0.5, 25.0, 5e-4, 100)
    assert len(hist) == ref["nsteps"] and ref["finished"]
    assert [h[1][0] for h in hist] == ref["etotal_all"]
    lv, en = hist[-1]
    assert lv == [x[1] for x in ref["steps"][-1]["levels"]]
    assert en == ref["steps"][-1]["energies"]
    # README.md:62-74 (six printed decimals)
    readme_levels = [-113.800134, -10.794172, -8.443439, -0.883384, -0.382330]
    readme_en = [-525.946200, 524.969813, 231.458124, -1253.131983, -29.242154]
    assert [round(x, 6) for x in lv] == readme_levels
    assert [round(x, 6) for x in en] == readme_en


def test_radon_reference_values_recorded():
    """The Radon (README.md:30-54) end-to-end reference run is recorded; the oracle is compared to it in
    tests/test_oracle_vs_ref.py::test_radon_first_step (slow) and the GPU path in the -m gpu suite."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rn_end_to_end.json")
    if not os.path.exists(path):
        pytest.skip("rn_end_to_end.json not generated")
    rn = json.load(open(path))
    last = rn["Rn_LDA_L17"]["last"]
    readme = [-3204.756288, -546.577961, -527.533025, -133.369145, -124.172863, -106.945007, -31.230804,
              -27.108985, -19.449995, -8.953318, -5.889683, -4.408703, -1.911330, -0.626571, -0.293180]
    assert [round(x[1], 6) for x in last["levels"]] == readme
    en = last["energies"]
    assert round(en[0], 6) == -21861.346900 and round(en[2], 6) == 8632.016044 and round(en[4], 6) == -381.915254
    # Ekin/Eenuc differ from the README in the 6th decimal across platforms
    assert abs(en[1] - 21854.672704) < 5e-6 and abs(en[3] + 51966.120394) < 5e-6


def test_hinted_brackets_reproduce_reference_scf(golden):
    """The batched GPU mode starts level k at E_{k-1}(previous SCF step) - 3 instead of E_{k-1}(this step) - 3
    (DFTAtom.cpp:541).  Restated in the oracle (chained == 2) and run to the reference's step count on the README
    Argon configuration: every step's Etotal within 1e-9 relative of the reference's (measured: < 7e-11, the SCF's own
    round-off jitter), converged eigenvalues within 1e-8 Ha and identical at the README's 6 decimals.
    The step at which `Finished!` fires is round-off noise and is not compared."""
    _, meta = golden
    ref = meta["end_to_end"]["Ar_LDA_L14"]
    hist = _run_scf(0, 18, 14, 0.5, 25.0, 5e-4, ref["nsteps"], chained=2)
    et = np.array([h[1][0] for h in hist])
    want = np.array(ref["etotal_all"][: len(et)])
    assert len(et) >= 25
    assert np.max(np.abs(et - want) / np.abs(want)) < 1e-9
    lv, en = hist[-1]
    assert np.max(np.abs(np.array(lv) - np.array([x[1] for x in ref["steps"][-1]["levels"]]))) < 1e-8
    assert [round(x, 6) for x in lv] == [-113.800134, -10.794172, -8.443439, -0.883384, -0.382330]   # README.md:62-74


def test_unchained_brackets_break_f_levels():
    """Why the batched mode is hinted and not simply un-chained: started from -Z^2-1, the node-count search of an
    f level (l=3) converges to the bracket bottom instead of the eigenvalue."""
    o = O.oracle()
    g = O.make_grid(12, 2e-3, 50.0)
    V = O.coulomb_potential(g, 86)
    lv = [(3, 3, 14)]                                   # 4f of a bare Z=86 Coulomb field: -Z^2/32 = -231.125
    for chained, bottom in ((0, -86.0 ** 2 - 1.0), (2, -234.0)):
        arr = O.levels_array(lv)
        nd = np.zeros(g.N)
        Eel, Bot = C.c_double(0), C.c_double(bottom)
        hints = np.array([bottom])
        o.dfo_loop_over_levels(C.byref(g), O.dp(V), arr, 1, O.dp(nd), C.byref(Eel), C.byref(Bot), chained, O.dp(hints))
        if chained == 0:
            assert abs(arr[0].E - bottom) < 1e-6          # stuck at the bottom of the bracket
        else:
            assert abs(arr[0].E + 231.125) < 1e-3


def test_reference_rounding_sensitivity_of_scf_steps(tmp_path):
    """How far apart are two correctly rounded executions of the reference's OWN algorithm after an SCF step?  The oracle
    (bit-identical to the compiled reference, tests/test_oracle_vs_ref.py) is built a second time with FMA contraction
    allowed -- same source, same algorithm, different roundings -- and both run two SCF steps of Ni on the 131073-node
    grid.  Step 0 starts from the same potential: eigenvalues agree to ~1e-10 Ha.  Step 1 sees the potential that step 0's
    Poisson solve produced, whose 100-V-cycle end state is round-off noise amplified through 1/r: eigenvalues differ by
    ~1e-8 Ha (1e-10 relative and more).  This is the floor under every "SCF step k >= 1 vs the reference" comparison of
    the GPU suite (tests/test_gpu_configs.py: gate 1e-8 Ha + 2e-9 |E| for step 1)."""
    import ctypes as C
    import subprocess
    so = str(tmp_path / "libdfta_oracle_fma.so")
    src = os.path.join(O.ORACLE_DIR, "dfta_oracle.c")
    r = subprocess.run(["gcc", "-O2", "-std=c11", "-ffp-contract=fast", "-mfma", "-fPIC", "-shared", "-o", so, src, "-lm"], capture_output=True)
    if r.returncode != 0:
        pytest.skip("no FMA-capable build on this host")
    o = O.oracle()
    f = C.CDLL(so)
    for name in ("dfo_scf_create", "dfo_scf_destroy", "dfo_scf_step"):
        g, h = getattr(f, name), getattr(o, name)
        g.restype, g.argtypes = h.restype, h.argtypes
    L, d, R, Z = 17, 1e-4, 50.0, 28
    a, b = o.dfo_scf_create(0, Z, L, 0.5, R, d, 1), f.dfo_scf_create(0, Z, L, 0.5, R, d, 1)
    ea, eb = O.Energies(), O.Energies()
    diffs = []
    for _ in range(2):
        o.dfo_scf_step(a, C.byref(ea))
        f.dfo_scf_step(b, C.byref(eb))
        la = np.array([a.contents.la[i].E for i in range(a.contents.nla)])
        lb = np.array([b.contents.la[i].E for i in range(b.contents.nla)])
        diffs.append((np.max(np.abs(la - lb)), abs(ea.Etotal - eb.Etotal) / abs(ea.Etotal)))
    o.dfo_scf_destroy(a)
    f.dfo_scf_destroy(b)
    print("plain vs FMA build of the same algorithm, Ni @ 131073: step 0 max |dE| %.2e Ha, step 1 max |dE| %.2e Ha" % (diffs[0][0], diffs[1][0]))
    assert diffs[0][0] < 2e-9                      # same potential: only the sweeps' roundings differ
    assert 1e-9 < diffs[1][0] < 1e-6               # one Poisson solve later: the noise floor of the SCF itself
    assert diffs[1][1] < 1e-9                      # total energies stay inside the 1e-9 relative gate


def test_uniform_grid_oracle_vs_golden():
    """The oracle's uniform-grid restatement against tests/golden/uniform.npz (vectors of the compiled reference): sweeps,
    match, level driver and SolvePoissonUniform, bit for bit -- the pin that travels to machines without /root/reference."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    data = np.load(os.path.join(here, "uniform.npz"))
    meta = json.load(open(os.path.join(here, "uniform_meta.json")))
    o = O.oracle()
    m = meta["grid"]
    g = O.make_ugrid(m["L"], m["Rmax"])
    assert g.N == m["N"] and g.h == m["h"]
    rr = g.h * np.arange(g.N)
    pots = {"coulomb10": np.concatenate([[0.0], -10.0 / rr[1:]]), "screened18": screened_potential(rr, 18.0)}
    P = np.zeros(g.N)
    for pname, V in pots.items():
        rows = data["sweeps_" + pname]
        for l, E, lim, cnt, u0 in rows[::5]:
            assert o.dfo_ucount_nodes(C.byref(g), O.dp(V), int(l), float(E), int(lim), None) == int(cnt)
            got = o.dfo_usolution_in_zero(C.byref(g), O.dp(V), int(l), float(E))
            assert got == u0 or (np.isnan(got) and np.isnan(u0))
        for row in data["match_" + pname][::4]:
            mp = o.dfo_umatch(C.byref(g), O.dp(V), int(row[0]), float(row[1]), O.dp(P))
            assert mp == int(row[2]) and np.nansum(P) == row[3] and np.nansum(np.abs(P)) == row[4]
    V = pots["screened18"]
    lv = O.subshells(18)
    lev = O.levels_array(lv)
    nd = np.zeros(g.N)
    eel, bot = C.c_double(0), C.c_double(-18.0 * 18 - 1.0)
    conv = o.dfo_uloop_over_levels(C.byref(g), O.dp(V), lev, len(lv), O.dp(nd), C.byref(eel), C.byref(bot))
    assert [lev[k].E for k in range(len(lv))] == list(data["levels_E"])
    assert np.array_equal(nd[:: g.N // 256], data["levels_newdensity_sample"])
    assert [eel.value, bot.value, float(conv), nd.sum()] == list(data["levels_scalars"])
    pg = meta["poisson_grid"]
    gp = O.make_ugrid(pg["L"], pg["Rmax"])
    rp = gp.h * np.arange(gp.N)
    for tag, Z in (("Z2", 2), ("Z18", 18)):
        p = o.dfo_poisson_create(pg["L"], 0.0)
        U = np.zeros(gp.N)
        o.dfo_solve_poisson_uniform(p, Z, pg["Rmax"], O.dp(Z * np.exp(-2 * rp) / np.pi), O.dp(U))
        assert np.array_equal(U, data["poisson_%s_U" % tag])
        o.dfo_poisson_destroy(p)


def test_reference_late_step_jitter():
    """Why "converged" eigenvalues are compared at 2e-7 Ha (Rn), 1e-6 Ha (Z <= 86) and 8e-6 Ha (Z > 86) in tests/test_gpu_configs.py and not at
    BASELINE.md's 1e-8: the COMPILED REFERENCE's own eigenvalues still move by that much from one SCF step to the next when it stops (or hits
    its 100-step cap) -- the multigrid's 100-V-cycle end state is a round-off floor that the 1/r-weighted potential amplifies.  Fixture:
    tests/golden/late_step_jitter.json (make_golden_table.py jitter: the last four steps of Rn and of a sample of the loosest atoms, L = 17).
    Asserted: the jitter is there (>= 3e-8 Ha for every atom of the sample: 3.6e-8 for Ni ... 1.3e-6 for Gd), and every gate is within 4x (atoms that finish) / 12x (atoms that
    run to the cap, whose trajectories keep wandering) of the largest step-to-step movement observed in its group -- while Etotal, which is
    variational, is steady to 1e-9 relative."""
    import json
    import numpy as np
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "late_step_jitter.json")) as f:
        d = json.load(f)
    jit = {}
    for Z, rec in d.items():
        lv = [np.array([x[1] for x in s["levels"]]) for s in rec["last_steps"]]
        en = [s["energies"][0] for s in rec["last_steps"]]
        jit[int(Z)] = max(float(np.max(np.abs(lv[i + 1] - lv[i]))) for i in range(len(lv) - 1))
        assert jit[int(Z)] >= 3e-8, (Z, jit[int(Z)])
        assert max(abs((en[i + 1] - en[i]) / en[i]) for i in range(len(en) - 1)) <= 1e-9
    groups = {"Rn": ([86], 2e-7, 4.0), "Z <= 86": ([z for z in jit if z < 86], 1e-6, 4.0), "Z > 86": ([z for z in jit if z > 86], 8e-6, 12.0)}
    for name, (zs, gate, factor) in groups.items():
        worst = max(jit[z] for z in zs)
        print("%s: largest eigenvalue movement between the reference's last steps %.1e Ha, gate %.0e Ha" % (name, worst, gate))
        assert gate <= factor * worst, (name, worst, gate)
