"""GPU suite (-m gpu): the BASELINE.json configurations that round 1 left without a parity test.

  configs[2]  Rn LSDA @ 131073 nodes       vs tests/golden/rn_end_to_end.json (compiled reference: first, second, last step)
  configs[3]  Z = 1..86 LDA @ 131073 nodes vs tests/golden/periodic_table_L17.json (compiled reference, every atom run to
              the reference's own stop: first two steps and the stop state), as ONE batch on one GPU
  configs[4]  1 048 577 nodes (20 levels)  vs tests/golden/l20.npz / l20_meta.json: sweeps (bit-exact), one Poisson solve,
              the first two SCF steps of Rn LSDA -- at this size level 0 of the multigrid is not staged in LDS
  README      Rn table of the reference's README.md:30-52 through the headless front end (six printed decimals)

Tolerances, in BASELINE.md section 3's terms (printed maxima are the observed values):
  * node counts, cut-off indices, u(0), Psi with host boundary values: bit-exact;
  * per-step energies from the same start: 1e-9 relative; per-step eigenvalues: 1e-8 Ha + 1e-10 |E|
    (the Hartree potential carries the multigrid's round-off floor, which 1/r hands to the core levels);
  * SCF step k >= 1 (the potential has been through a Poisson solve): eigenvalues 1e-8 Ha + 2e-9 |E| -- the reference
    differs from ITSELF by that much when it is merely compiled with FMA contraction
    (tests/test_oracle_golden.py::test_reference_rounding_sensitivity_of_scf_steps); energies stay at 1e-9;
  * converged energies (closed shells, Rn): 1e-9 relative for Etotal; 2e-9 for the components (Ekin and Eenuc are
    differences of large terms and jitter by ~1e-9 between the reference's own late steps);
  * converged eigenvalues: 2e-7 Ha + 1e-10 |E|: BASELINE's 1e-8 Ha gate is RELAXED here because the reference's own
    eigenvalues move by up to ~1e-7 Ha between its last steps (its stop test looks at Etotal only, and the stop step is
    round-off noise) -- the two runs stop at different steps of the same jitter;
  * final states over the whole periodic table (open shells included): Etotal 3e-9 (1e-9 where both runs met the stop
    test), components 2e-8, eigenvalues 1e-6 Ha: see test_periodic_table_batch_vs_reference.
"""
import json
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import dftatom_amd as D
from _knobs import knobs                 # noqa: E402
from golden.make_golden import GRIDS, screened_potential   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def ctx(torch_first):
    c = D.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def grid17(ctx):
    L, d, R = GRIDS["L17"]
    g = D.Grid(ctx, L, d, R)
    yield g
    g.close()


def _levels_of(scf, atom, lsda):
    return np.concatenate([scf.levels(atom, 0)["E"]] + ([scf.levels(atom, 1)["E"]] if lsda else []))


def _check_step(scf, atom, lsda, want, tag, stats, lv_rel=1e-10, en_abs_rel_etot=0.0):
    """one SCF step's printed values: energies 1e-9 relative, eigenvalues 1e-8 Ha + lv_rel |E|.  en_abs_rel_etot (tolerance modes): an
    energy COMPONENT may also differ by that fraction of |Etotal| -- the scan sweeps' eigenvalues are within 6e-11 |E| + 6e-10 Ha of the
    exact search (tests/test_gpu_scan.py), and a component that is a small difference of large terms (hydrogen's kinetic term in the
    first step: -0.0097 Ha of -0.545) turns 1e-11 Ha into 3e-9 of itself."""
    en, _ = scf.energies()
    want_lv = np.array([x[1] for x in want["levels"]])
    got_lv = _levels_of(scf, atom, lsda)
    assert len(got_lv) == len(want_lv), tag
    dlv = np.abs(got_lv - want_lv)
    den = np.array([abs(a - b) / abs(b) for a, b in zip(en[atom].as_list(), want["energies"])])
    stats["lv"] = max(stats.get("lv", 0.0), float(np.max(dlv - lv_rel * np.abs(want_lv))))
    stats["lvrel"] = max(stats.get("lvrel", 0.0), float(np.max(dlv / np.abs(want_lv))))
    stats["en"] = max(stats.get("en", 0.0), float(den.max()))
    if os.environ.get("DFTA_TEST_VERBOSE"):
        print(tag, "dE levels", np.array2string(dlv, precision=2), "energies rel", np.array2string(den, precision=2))
    assert np.all(dlv <= 1e-8 + lv_rel * np.abs(want_lv)), (tag, dlv.max())
    slack = en_abs_rel_etot * abs(want["energies"][0])
    assert all(abs(a - b) <= 1e-9 * abs(b) + slack for a, b in zip(en[atom].as_list(), want["energies"])), (tag, den)


def _check_converged(en, lv, want, tag, stats, lv_abs=2e-7, comp_rel=2e-9):
    want_lv = np.array([x[1] for x in want["levels"]])
    dlv = np.abs(lv - want_lv)
    den = np.array([abs(a - b) / abs(b) for a, b in zip(en, want["energies"])])
    stats["lv"] = max(stats.get("lv", 0.0), float(np.max(dlv - 1e-10 * np.abs(want_lv))))
    stats["etot"] = max(stats.get("etot", 0.0), float(den[0]))
    stats["comp"] = max(stats.get("comp", 0.0), float(den[1:].max()))
    if lv_abs is None:
        return
    assert np.all(dlv <= lv_abs + 1e-10 * np.abs(want_lv)), (tag, dlv.max())
    assert den[0] <= 1e-9 and np.all(den[1:] <= comp_rel), (tag, den)


# ---------------------------------------------------------------------------------------------------------------
# configs[2]: Rn LSDA @ 131073
# ---------------------------------------------------------------------------------------------------------------
def test_radon_lsda_vs_reference(ctx, grid17):
    rn = json.load(open(os.path.join(HERE, "golden", "rn_end_to_end.json")))["Rn_LSDA_L17"]
    stats = {}
    scf = D.Scf(ctx, grid17, [86], lsda=True, levels_mode=D.LEVELS_CHAINED)     # the reference's bracket hand-over
    for key in ("first", "second"):
        st = scf.step()
        _check_step(scf, 0, True, rn[key], key, stats)
        assert st.vcycles == 100 and st.sweeps_reference > 4000                  # 30 levels: twice the LDA batch
    scf.close()
    print("Rn LSDA per-step: max eigenvalue excess over 1e-10|E| %.2e Ha, max energy %.2e rel" % (stats["lv"], stats["en"]))
    # the product's default path for the number of steps the reference took; every step's Etotal against its trajectory
    want = np.array(rn["etotal_all"])
    scf = D.Scf(ctx, grid17, [86], lsda=True)
    traj = []
    for _ in range(len(want)):
        scf.step(want_stats=False)
        traj.append(scf.energies()[0][0].Etotal)
    traj = np.array(traj)
    rel = np.abs(traj - want) / np.abs(want)
    print("Rn LSDA trajectory (%d steps): max |dEtotal|/|Etotal| %.2e" % (len(want), rel.max()))
    assert rel.max() <= 1e-9
    cs = {}
    _check_converged(scf.energies()[0][0].as_list(), _levels_of(scf, 0, True), rn["last"], "Rn LSDA", cs)
    print("Rn LSDA converged: eigenvalue excess %.2e Ha, Etotal %.2e, components %.2e" % (cs["lv"], cs["etot"], cs["comp"]))
    # closed shells: alpha and beta levels coincide
    a, b = scf.levels(0, 0)["E"], scf.levels(0, 1)["E"]
    assert np.max(np.abs(a - b)) <= 1e-9
    scf.close()


# ---------------------------------------------------------------------------------------------------------------
# configs[3]: the periodic table as one batch
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("modes", ["exact", "tolerance", "adaptive"])
def test_periodic_table_batch_vs_reference(ctx, grid17, modes):
    """(modes = "tolerance": the same run with the scan sweeps and the multigrid's tolerance mode -- the opt-in fast path has to hold the
    same gates against the compiled reference for every atom of the table, not only for Rn.)
    Z = 1..86 as ONE batch on one GPU (the unit that examples/periodic_table.py shards over ranks), default product path,
    every atom advanced until it meets the reference's stop test or the reference's cap of 100 steps (DFTAtom.cpp:396):
      * steps 0 and 1 of every atom against the compiled reference's first two steps.  Step 0 (identical start potential):
        eigenvalues 1e-8 Ha + 1e-10 |E|.  Step 1 sees the potential that step 0's Poisson solve left, and the 100-V-cycle end
        state of that solve is round-off noise at the 1e-10 Z level which 1/r hands to every level: the reference differs FROM
        ITSELF there by 1.4e-8 Ha (Ni) / 9.2e-8 Ha (Te) when it is merely compiled with FMA contraction
        (tests/test_oracle_golden.py::test_reference_rounding_sensitivity_of_scf_steps), so the gate for step 1 is
        1e-8 Ha + 2e-9 |E| (observed maximum 1.4e-9 |E|).  Energies: 1e-9 relative in both steps;
      * every atom's final state against the reference's final state (converged gates).  The stop step itself is round-off
        noise (34 / 33 / 37 for three builds of the reference; 32 of the 86 atoms never trip the reference's
        |dE/E| < 1e-11 test within 100 steps although their energies have long settled), so step counts and Finished flags
        are reported, not compared."""
    table = json.load(open(os.path.join(HERE, "golden", "periodic_table_L17.json")))
    have = set(int(z) for z in table)
    Zs = []
    while len(Zs) + 1 in have:
        Zs.append(len(Zs) + 1)
    # Z = 1..86 is BASELINE config 4; round 3 extended the fixture to the front end's full range Z <= 118 (OptionsFrame.cpp:153;
    # Ac/Th/Pa/U/Np/Cm/Lr exceptions of AufbauPrinciple.h:101-117)
    assert len(Zs) >= 86, "tests/golden/periodic_table_L17.json must hold at least Z = 1..86"
    per_step, conv = {}, {}
    # steps 0 and 1 on the reference's own bisection path (bracket hand-over from level to level, DFTAtom.cpp:541)
    kw = {}
    if modes != "exact":            # "adaptive": the V-cycles stop on the round-off floor (DFTA_POISSON_ADAPTIVE); gated like "tolerance"
        kw = dict(sweep_mode=D.SWEEPS_TOLERANCE, poisson_mode=D.POISSON_ADAPTIVE if modes == "adaptive" else D.POISSON_TOLERANCE)
    scf = D.Scf(ctx, grid17, Zs, lsda=False, levels_mode=D.LEVELS_CHAINED, **kw)
    for it in range(2):
        scf.step(want_stats=False)
        for k, z in enumerate(Zs):
            _check_step(scf, k, False, table[str(z)]["first" if it == 0 else "second"], "Z=%d step %d" % (z, it), per_step,
                        lv_rel=1e-10 if it == 0 else 2e-9, en_abs_rel_etot=2e-10 if modes != "exact" else 0.0)
    scf.close()
    # the product's default path to the end
    scf = D.Scf(ctx, grid17, Zs, lsda=False, **kw)
    cap = 100
    nsteps = 0
    # (tolerance run) every atom's own movement in its last step: eigenvalues (Ha) and energy components (relative)
    track = modes != "exact"
    last_lv, last_en, move_lv, move_en = {}, {}, {}, {}
    live = np.ones(len(Zs), bool)
    for it in range(cap):
        scf.step(want_stats=False)
        nsteps += 1
        en_now, fin = scf.energies()
        if track:
            for k in np.nonzero(live)[0]:
                lv_k, en_k = np.array(scf.levels(int(k), 0)["E"]), np.array(en_now[k].as_list())
                if k in last_lv:
                    move_lv[k] = float(np.max(np.abs(lv_k - last_lv[k])))
                    move_en[k] = float(np.max(np.abs(en_k[1:] - last_en[k][1:]) / np.abs(en_k[1:])))
                last_lv[k], last_en[k] = lv_k, en_k
            live = ~fin.astype(bool)
        if fin.all():
            break
    en, fin = scf.energies()
    worst = []
    for k, z in enumerate(Zs):
        ref = table[str(z)]
        c = {}
        _check_converged(en[k].as_list(), scf.levels(k, 0)["E"], ref["last"], "Z=%d" % z, c, lv_abs=None)
        # tolerance run: what exceeds four of the atom's own last steps is what counts (see the gates below)
        worst.append((c["etot"], max(0.0, c["comp"] - 4 * move_en.get(k, 0.0)), max(0.0, c["lv"] - 4 * move_lv.get(k, 0.0)), z,
                      bool(ref["finished"] and fin[k])))
    # Final-state gates.  Both runs end where Etotal has stopped moving (or at the cap); Etotal is variational (second order in
    # what is left of the density error), its components and the eigenvalues are first order, and the two runs end at different
    # steps of the same jitter.  Gates = about twice the observed maxima (asserted below, printed with the summary):
    #   Z = 1..86   (BASELINE config 4): Etotal 3e-9 relative (1e-9 where both runs met the stop test; observed 1.8e-9 / 6e-10),
    #               components 2e-8 (1.1e-8), eigenvalues 1e-6 Ha + 1e-10 |E| (4.9e-7)
    #   Z = 87..118 (round 3; ten of these atoms never meet the reference's stop test within its 100 steps and still move by
    #               ~1e-6 Ha per step at the cap): Etotal 3e-9, components 3e-8 (1.2e-8), eigenvalues 8e-6 Ha (3.7e-6, Z = 111)
    # Tolerance modes: the stop test |dE/E| < 1e-11 is decided by round-off (late-step jitter 5e-10), so a different order of roundings
    # stops an atom at a different coincidence, and an atom that runs to the cap is caught at a different phase of its wandering --
    # observed: Ce (Z = 58) meets the test around step 22 in this batch (step 42 when run alone, 60 in exact mode, 65 in the reference; the
    # single-atom trajectories agree step by step to 1.6e-7 Ha) where its eigenvalues still move by 2e-6 Ha per step, and ends 3.7e-6 Ha /
    # 3.6e-8 from the reference's end state; Z > 86: 1.1e-5 Ha.  The tolerance run is therefore gated on what EXCEEDS four of the atom's
    # own last step (eigenvalues in Ha, components relative); Etotal, which is variational, with the exact mode's bounds.
    rows = ((1, 86, 3e-9, 1e-9, 2e-8, 1e-6), (87, 118, 3e-9, 1e-9, 3e-8, 8e-6))
    if modes != "exact":
        rows = ((1, 86, 3e-9, 1e-9, 1e-8, 5e-7), (87, 118, 3e-9, 1e-9, 1e-8, 5e-7))       # observed: Etotal 8.8e-10 / 4.1e-10, excess 2.2e-9, 2.2e-8 Ha (adaptive: 1.7e-7 Ha)
    for lo, hi, g_et, g_etf, g_comp, g_lv in rows:
        grp = [w for w in worst if lo <= w[3] <= hi]
        if not grp:
            continue
        m_et = max(w[0] for w in grp)
        m_etf = max([w[0] for w in grp if w[4]] or [0.0])
        m_comp = max(w[1] for w in grp)
        m_lv = max(w[2] for w in grp)
        print("  final states Z = %d..%d: Etotal %.2e (both finished: %.2e), components %.2e, eigenvalue excess %.2e Ha; worst eigenvalue at Z=%d"
              % (lo, min(hi, Zs[-1]), m_et, m_etf, m_comp, m_lv, max(grp, key=lambda w: w[2])[3]))
        assert m_et <= g_et and m_etf <= g_etf, (lo, hi, m_et, m_etf)
        assert m_comp <= g_comp and m_lv <= g_lv, (lo, hi, m_comp, m_lv)
    for c_et, c_comp, c_lv, z, _ in worst:
        for key, v in (("lv", c_lv), ("etot", c_et), ("comp", c_comp)):
            conv[key] = max(conv.get(key, 0.0), v)
    fin_ref = np.array([table[str(z)]["finished"] for z in Zs])
    print("periodic table: %d atoms in one batch, %d steps of the batch; Finished here %d, in the reference %d (both %d)"
          % (len(Zs), nsteps, int(fin.sum()), int(fin_ref.sum()), int((fin_ref & fin.astype(bool)).sum())))
    print("  first two steps: max |dE_level| / |E| %.2e, energies %.2e rel" % (per_step["lvrel"], per_step["en"]))
    print("  final states   : eigenvalue excess %.2e Ha, Etotal %.2e, components %.2e; worst Etotal at Z=%d, worst component at Z=%d"
          % (conv["lv"], conv["etot"], conv["comp"], max(worst)[3], max(worst, key=lambda w: w[1])[3]))
    if os.environ.get("DFTA_TEST_VERBOSE"):
        for w in sorted(worst, reverse=True)[:12]:
            print("   Z=%2d  Etotal %.2e  components %.2e  eigenvalues %.2e  (finished here %d, reference %d after %d steps)"
                  % (w[3], w[0], w[1], w[2], fin[w[3] - 1], table[str(w[3])]["finished"], table[str(w[3])]["nsteps"]))
    scf.close()


# ---------------------------------------------------------------------------------------------------------------
# configs[4]: 1 048 577 nodes
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def grid20(ctx):
    L, d, R = GRIDS["L20"]
    g = D.Grid(ctx, L, d, R)
    yield g
    g.close()


def test_l20_sweeps_bit_exact(ctx, grid20):
    data = np.load(os.path.join(HERE, "golden", "l20.npz"))
    V = screened_potential(grid20.r(), 86.0)
    rows = data["sweeps"]                                  # l, E, limit, count, u0, cut-off
    for which in (D.SWEEP_KERNEL_FUSED, D.SWEEP_KERNEL_PIPELINED):
        ctx.set_sweep_kernel(which)
        c = D.numerov_sweeps(ctx, grid20, D.SWEEP_COUNT, V, rows[:, 0], rows[:, 1], rows[:, 2])
        assert np.array_equal(c["count"], rows[:, 3].astype(np.int32))
        assert np.array_equal(c["start"], rows[:, 5].astype(np.int32))
        z = D.numerov_sweeps(ctx, grid20, D.SWEEP_ZERO, V, rows[:, 0], rows[:, 1])
        assert np.array_equal(z["u0"].view(np.int64), rows[:, 4].copy().view(np.int64))
    ctx.set_sweep_kernel(D.SWEEP_KERNEL_AUTO)
    m = data["match"]
    psi, mp = D.numerov_match(ctx, grid20, V, m[:, 0], m[:, 1])
    assert np.array_equal(mp, m[:, 2].astype(np.int64))
    assert np.array_equal(np.nansum(psi, axis=1), m[:, 3]) and np.array_equal(np.nansum(np.abs(psi), axis=1), m[:, 4])
    assert np.array_equal(psi[:, :: grid20.N // 64][:, :64], m[:, 5:])


def test_l20_poisson_vs_reference(ctx, grid20):
    data = np.load(os.path.join(HERE, "golden", "l20.npz"))
    rr = grid20.r()
    rho = 86.0 * np.exp(-2 * rr) / np.pi
    ps = D.Poisson(ctx, grid20, 1)
    U, vc, err = ps.solve([86], rho)
    ps.close()
    want = data["poisson_U_sample"]
    got = U[0][::64]
    print("L20 Poisson: max |dU| %.2e (Z = 86), V-cycles %d" % (np.max(np.abs(got - want)), vc[0]))
    assert np.max(np.abs(got - want)) <= 1e-10 * 86
    cs = data["poisson_U_checksum"]
    assert abs(U[0].sum() - cs[0]) <= 1e-11 * abs(cs[0])
    assert vc[0] == 100
    # staged / unstaged and group variants agree bit for bit at this size too
    ref = U[0].view(np.int64)
    for var in ({"DFTA_POISSON_GROUP": "0"}, {"DFTA_POISSON_NOSTAGE": "1"}, {"DFTA_POISSON_GROUP": "3"}, {"DFTA_POISSON_GROUP": "2"}):
        with knobs(var):
            p2 = D.Poisson(ctx, grid20, 1)
            U2, _, _ = p2.solve([86], rho)
            p2.close()
        assert np.array_equal(U2[0].view(np.int64), ref), var


@pytest.mark.parametrize("modes", ["exact", "tolerance"])
def test_l20_radon_lsda_steps_vs_reference(ctx, grid20, modes):
    """First two SCF steps of Rn LSDA at 1 048 577 nodes against the compiled reference (modes = "tolerance": with the scan sweeps -- 2048
    rows per lane -- and the multigrid's tolerance mode; same gates, a component may also differ by 2e-10 |Etotal|).

    Step 0 (identical start potential): the usual per-step gates.  From step 1 on the comparison is limited by the
    CONDITIONING of the reference's own Poisson solve at this size, not by this implementation: the solver here returns
    the reference's U bit for bit when it is given the reference's density (test_l20_poisson_vs_reference), but the
    100-V-cycle end state amplifies a perturbation of the density in the twelfth digit -- which is what the device's
    exp() in the sweeps' start values makes of the step-0 density -- to ~1e-9 of U (second differences of U over 2^20
    nodes cancel ten digits).  The test measures that amplification on the step-0 density and gates step 1 with it."""
    meta = json.load(open(os.path.join(HERE, "golden", "l20_meta.json")))
    steps = meta["Rn_LSDA_L20"]["steps"]
    kw = dict(sweep_mode=D.SWEEPS_TOLERANCE, poisson_mode=D.POISSON_TOLERANCE) if modes == "tolerance" else {}
    scf = D.Scf(ctx, grid20, [86], lsda=True, levels_mode=D.LEVELS_CHAINED, **kw)
    stats = {}
    st = scf.step()
    # tolerance: at this size the REFERENCE's double recurrence is up to 1.3e-9 |E| away from the 80-bit eigenvalue (7.8e-11 ... 1.3e-9 from
    # level to level; the scan's summed form 7e-13: tests/test_scan_precision.py, CPU), so the scan sweeps are gated at 1e-9 |E| here
    _check_step(scf, 0, True, steps[0], "L20 step 0", stats, lv_rel=1e-9 if modes == "tolerance" else 1e-10,
                en_abs_rel_etot=2e-10 if modes == "tolerance" else 0.0)
    assert st.vcycles == 100
    if modes == "tolerance":
        assert st.levels_layout == 4                     # the scan search ran (no hand-back to the exact kernels)
    print("Rn LSDA @ 1048577, step 0: eigenvalue excess over 1e-10|E| %.2e Ha, energies %.2e rel" % (stats["lv"], stats["en"]))
    rho = scf.array(0)
    ps = D.Poisson(ctx, grid20, 1)
    rng = np.random.default_rng(5)
    Ua, _, _ = ps.solve([86], rho)
    Ub, _, _ = ps.solve([86], rho * (1.0 + 1e-12 * rng.standard_normal(rho.size)))
    ps.close()
    kabs = float(np.max(np.abs(Ua - Ub)))
    krel = float(np.max(np.abs(Ua[0, 1:] - Ub[0, 1:]) / np.abs(Ua[0, 1:])))
    # the gate of step 1 is the conditioning measured ON THE REFERENCE (tests/golden/make_golden_table.py l20cond: the compiled reference's
    # own solve moves by max |dU| = 5.4e-7 under the same 1e-12 perturbation of a Z = 86 density) -- VERDICT r3: not the product's own
    # number, which is only required to be of that size
    cond = meta["poisson_conditioning"]
    ref_abs, ref_rel = cond["max_abs_dU"], cond["max_rel_dU"]
    print("  conditioning of the solve: density perturbed by 1e-12 relative -> max |dU| %.2e (%.2e relative); the reference: %.2e (%.2e)"
          % (kabs, krel, ref_abs, ref_rel))
    assert ref_abs > 1e-8                                # twelve digits in, fewer than nine out
    assert ref_abs / 8 <= kabs <= 8 * ref_abs            # the product's solver is conditioned like the reference's
    st = scf.step()
    en, _ = scf.energies()
    want_lv = np.array([x[1] for x in steps[1]["levels"]])
    dlv = np.abs(_levels_of(scf, 0, True) - want_lv)
    den = np.array([abs(a - b) / abs(b) for a, b in zip(en[0].as_list(), steps[1]["energies"])])
    print("  step 1: max eigenvalue difference %.2e Ha, energies %.2e rel (gates: %.1e Ha, %.1e rel)"
          % (dlv.max(), den.max(), 4 * ref_abs + 1e-8, 4 * ref_rel + 1e-9))
    tol_rel = 1e-9 if modes == "tolerance" else 0.0          # the reference's rounding bias at this size (see step 0)
    assert np.all(dlv <= 4 * ref_abs + 1e-8 + tol_rel * np.abs(want_lv)) and den.max() <= 4 * ref_rel + 1e-9 + 2e-10 * (modes == "tolerance")
    assert st.vcycles == 100
    scf.close()


def test_l20_batch_layouts_agree(ctx, grid20):
    """BASELINE config 5's batch-of-atoms form (bench.py: rn_lsda_l20_batch16): 8 Rn atoms LSDA at 1 048 577 nodes -- 240 jobs: the
    device-side level search with one workgroup per level (+ 16 second ones; round 5: packed host rounds, LEVELS_PERSIST_WIDE=64), 16
    multigrid workgroups per atom -- two SCF steps; every copy of the atom gets the single atom's energies and eigenvalues bit for bit,
    under both layouts, and the layout reports itself in the step statistics."""
    one = D.Scf(ctx, grid20, [86], lsda=True)
    for _ in range(2):
        st1 = one.step()
    ref_e = one.energies()[0][0].as_list()
    ref_lv = [one.levels(0, sp)["E"].copy() for sp in range(2)]
    assert int(st1.levels_layout) == 5               # one atom: the device-side search (persist.inc)
    one.close()
    for knobs, layout in (("", 5), ("LEVELS_PERSIST_WIDE=64", 2)):
        old = os.environ.get("DFTA_DEBUG")
        if knobs:
            os.environ["DFTA_DEBUG"] = knobs
        try:
            b = D.Scf(ctx, grid20, [86] * 8, lsda=True)
        finally:
            if knobs:
                if old is None:
                    os.environ.pop("DFTA_DEBUG", None)
                else:
                    os.environ["DFTA_DEBUG"] = old
        for _ in range(2):
            st = b.step()
        print("8 x Rn LSDA @ 2^20+1, layout %d: level phase %.1f ms, %d rounds" % (int(st.levels_layout), st.ms_levels, int(st.rounds)))
        assert int(st.levels_layout) == layout and b.njobs == 240, (knobs, int(st.levels_layout))
        en, _ = b.energies()
        for a in range(8):
            assert en[a].as_list() == ref_e, (knobs, a)
            for sp in range(2):
                assert np.array_equal(b.levels(a, sp)["E"].view(np.int64), ref_lv[sp].view(np.int64)), (knobs, a, sp)
        b.close()


def test_l20_third_bisection_on_its_fixed_point(ctx, grid20):
    """From the seventh SCF step on, one of Rn's levels at 1 048 577 nodes ends its third bisection with |u(0)| >= 1e15: the reference
    (DFTAtom.cpp:517-534) then keeps halving an interval that has collapsed until its 500-iteration cap -- the same midpoint, the same
    sweep, the same decision ~440 times.  The level solver takes those iterations without integrating them once a step leaves the
    interval unchanged: same energies, eigenvalues, convergence flags and reference-equivalent sweep counts as with every iteration
    integrated (LEVELS_NOFIXEDPOINT), in 8-12 rounds instead of 70."""
    def run(knobs):
        old = os.environ.get("DFTA_DEBUG")
        if knobs:
            os.environ["DFTA_DEBUG"] = knobs
        try:
            scf = D.Scf(ctx, grid20, [86], lsda=True)
        finally:
            if knobs:
                if old is None:
                    os.environ.pop("DFTA_DEBUG", None)
                else:
                    os.environ["DFTA_DEBUG"] = old
        rows = []
        for _ in range(8):
            st = scf.step()
            rows.append((scf.energies()[0][0].as_list(), [scf.levels(0, sp)["E"].copy() for sp in range(2)],
                         [scf.levels(0, sp)["converged"].copy() for sp in range(2)], int(st.sweeps_reference), int(st.sweeps_reference_executed),
                         int(st.rounds), [scf.levels(0, sp)["status"].copy() for sp in range(2)]))
        scf.close()
        return rows
    got, full = run(""), run("LEVELS_NOFIXEDPOINT")
    hit = False
    for k, (x, y) in enumerate(zip(got, full)):
        assert x[0] == y[0] and x[3] == y[3], k
        for sp in range(2):
            assert np.array_equal(x[1][sp].view(np.int64), y[1][sp].view(np.int64)) and np.array_equal(x[2][sp], y[2][sp]), (k, sp)
        assert x[5] <= 20, (k, x[5])
        if y[5] >= 50:                                   # a step in which a level runs to the iteration cap
            hit = True
            assert x[4] <= y[4] - 300 and not (x[2][0].all() and x[2][1].all()), (k, x[4], y[4])
            # dfta_level_result.status tells the ways of not converging apart (the reference has one flag): the capped level stood on a
            # fixed point of its bisection; with every iteration integrated it is "iteration cap" alone -- the same level either way
            for sp in range(2):
                capped = ~x[2][sp].astype(bool)
                assert np.all(x[6][sp][capped] & D.LEVEL_ITERATION_CAP) and np.all(x[6][sp][capped] & D.LEVEL_FIXED_POINT)
                assert np.all(x[6][sp][~capped] == D.LEVEL_CONVERGED)
                assert np.all((y[6][sp][capped] & (D.LEVEL_ITERATION_CAP | D.LEVEL_FIXED_POINT)) == D.LEVEL_ITERATION_CAP)
    assert hit, [r[5] for r in full]


# ---------------------------------------------------------------------------------------------------------------
# README.md:30-52
# ---------------------------------------------------------------------------------------------------------------
def test_headless_front_end_reproduces_readme_radon():
    exe = os.path.join(ROOT, "dftatom_amd", "compat", "dftatom_cli")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe)])
    out = subprocess.run([exe, "86", "17", "0.5", "50", "0.0001", "0"], check=True, capture_output=True, text=True,
                         timeout=600).stdout
    assert "Finished!" in out
    lines = out.strip().splitlines()
    assert lines[-1].strip() == "1s2 2s2 2p6 3s2 3p6 3d10 4s2 4p6 4d10 4f14 5s2 5p6 5d10 6s2 6p6"
    last = [ln for ln in lines if ln.startswith("Energy")][-15:]
    got = [(re.search(r"Energy (\w+):", ln).group(1), float(re.search(r": (\S+) Num", ln).group(1)),
            int(ln.split("Num nodes: ")[1])) for ln in last]
    want = [("1s", -3204.756288, 0), ("2s", -546.577961, 1), ("2p", -527.533025, 0), ("3s", -133.369145, 2),
            ("3p", -124.172863, 1), ("3d", -106.945007, 0), ("4s", -31.230804, 3), ("4p", -27.108985, 2),
            ("4d", -19.449995, 1), ("4f", -8.953318, 0), ("5s", -5.889683, 4), ("5p", -4.408703, 3),
            ("5d", -1.911330, 2), ("6s", -0.626571, 5), ("6p", -0.293180, 4)]                     # README.md:32-46
    assert got == want
    et = [ln for ln in lines if ln.startswith("Etotal")][-1]
    vals = [float(x) for x in re.findall(r"= (-?\d+\.\d+)", et)]
    ref = [-21861.346900, 21854.672704, 8632.016044, -51966.120394, -381.915254]                   # README.md:47
    # Ekin / Eenuc differ in the sixth decimal between builds of the reference itself (...707 vs ...704):
    # 1e-9 relative, i.e. the fifth decimal of these five-digit numbers
    assert all(abs(a - b) <= 1e-9 * abs(b) + 1e-6 for a, b in zip(vals, ref)), (vals, ref)
