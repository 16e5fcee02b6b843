"""GPU suite (-m gpu): the TOLERANCE MODE of the Numerov sweeps (dftatom_amd/csrc/scan.hip: the recurrence of Numerov.h:309-321 as a
transfer-matrix scan, one workgroup per trial; k_scan_levels: the three bisections of DFTAtom.cpp:493-604 by one workgroup) against the
oracle, the golden vectors of the compiled reference and the exact kernels.

Gates (fp64; the order of roundings differs from the reference's, nothing else):
  * cut-off indices, classical-turning-point exits (loop trips): EXACT;
  * node counts (CountNodes' decision value min(count, limit + 1)): EXACT on every golden row, every ragged random row and every
    full-size row -- a count can differ only inside the round-off band of a transition, a few 1e-12 |E| wide, which no row hits;
  * u(0): 1e-8 relative away from its zeros (median ~1e-12);
  * per-level eigenvalues from identical V: 6e-11 |E| + 6e-10 Ha against the exact path -- that difference is the rounding BIAS of the
    reference's own recurrence (w only: every step subtracts two nearly equal numbers, the slope carries ~1e-10 relative error after
    131 073 steps), not of the scan: tests/test_scan_precision.py (CPU suite) shows the summed form (w and D = w - w', what scan.hip
    integrates) 1000x closer to the 80-bit eigenvalue than the reference's double arithmetic; sweep counts within 2 %;
  * SCF: energies 1e-9 relative, eigenvalues 1e-8 Ha + 2e-9 |E| against the compiled reference's goldens, Etotal 2e-9 along the whole
    recorded trajectory -- the gates of the multigrid's tolerance mode (tests/test_gpu_resident.py).
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import _oracle as O                     # noqa: E402  (checker only)
import dftatom_amd as D                 # noqa: E402
from golden.make_golden import GRIDS, screened_potential   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ctx(torch_first):
    c = D.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def grid14(ctx):
    L, d, R = GRIDS["L14"]
    g = D.Grid(ctx, L, d, R)
    yield g
    g.close()


@pytest.fixture(scope="module")
def grid17(ctx):
    L, d, R = GRIDS["L17"]
    g = D.Grid(ctx, L, d, R)
    yield g
    g.close()


def _pots(grid):
    rr = grid.r()
    V = np.zeros(grid.N)
    V[1:] = -18.0 / rr[1:]
    return {"coulomb18": V, "screened18": screened_potential(rr, 18.0), "screened86": screened_potential(rr, 86.0)}


@pytest.mark.parametrize("pname", ["coulomb18", "screened18", "screened86"])
def test_scan_sweeps_vs_golden(ctx, grid14, golden, pname):
    data, _ = golden
    V = _pots(grid14)[pname]
    rows = data[f"numerov_{pname}_counts"]
    res = D.numerov_sweeps_scan(ctx, grid14, D.SWEEP_COUNT, V, rows[:, 0], rows[:, 1], rows[:, 2])
    assert np.array_equal(res["count"], rows[:, 3].astype(np.int32))
    assert not res["fallback"].any()
    sw = data[f"numerov_{pname}_sweeps"]
    z = D.numerov_sweeps_scan(ctx, grid14, D.SWEEP_ZERO, V, sw[:, 0], sw[:, 1])
    assert np.array_equal(z["start"], sw[:, 3].astype(np.int32))
    rel = np.abs(z["u0"] - sw[:, 2]) / np.abs(sw[:, 2])
    rel = rel[np.isfinite(rel)]
    # the Coulomb rows include the hydrogenic eigenvalues themselves, where u(0) is round-off in either arithmetic (4 % of its rows)
    assert np.median(rel) <= 1e-11 and np.mean(rel > 1e-8) <= (0.05 if pname == "coulomb18" else 0.01), (np.median(rel), np.sort(rel)[-5:])


def test_scan_sweeps_ragged_vs_oracle(ctx, grid14):
    """random (potential, l, E, limit): counts, cut-offs and exit points equal the oracle's; positive and tiny energies, every limit"""
    o = O.oracle()
    g = O.make_grid(*GRIDS["L14"])
    P = _pots(grid14)
    V = np.stack([P["screened18"], P["screened86"]])
    rng = np.random.default_rng(4242)
    for nt in (1, 65, 1500):
        vidx = rng.integers(0, 2, nt).astype(np.int32)
        l = rng.integers(0, 4, nt).astype(np.int32)
        E = np.where(rng.random(nt) < 0.1, rng.uniform(0, 50, nt), -10.0 ** rng.uniform(-4, 3.8, nt))
        lim = rng.integers(0, 6, nt).astype(np.int32)
        c = D.numerov_sweeps_scan(ctx, grid14, D.SWEEP_COUNT, V, l, E, lim, vidx=vidx)
        z = D.numerov_sweeps_scan(ctx, grid14, D.SWEEP_ZERO, V, l, E, vidx=vidx)
        worst = 0.0
        for k in range(nt):
            st, tr = C.c_long(), C.c_long()
            want = o.dfo_count_nodes(C.byref(g), O.dp(V[vidx[k]]), int(l[k]), float(E[k]), int(lim[k]), C.byref(st), C.byref(tr))
            assert c["count"][k] == want and c["start"][k] == st.value, (nt, k, want, c["count"][k])
            if want <= lim[k]:
                assert c["trip"][k] == tr.value, (nt, k)       # no early return at count > limit: the trips are the turning-point exit's
            u0 = o.dfo_solution_in_zero(C.byref(g), O.dp(V[vidx[k]]), int(l[k]), float(E[k]), None)
            if np.isfinite(u0) and u0 != 0:
                worst = max(worst, abs(z["u0"][k] - u0) / abs(u0))
        assert worst <= 1e-7, worst
    assert D.numerov_sweeps_scan(ctx, grid14, D.SWEEP_ZERO, V, [], [])["u0"].size == 0      # empty batch


def test_scan_sweeps_pathological_potentials(ctx, grid14):
    """potentials no SCF produces: where the scan cannot decide (non-finite values, f >= 12 in a step row) it must SAY so (fallback
    flag: the level solver then hands the solve to the exact kernels); where it does decide, counts and cut-offs are the oracle's"""
    o = O.oracle()
    g = O.make_grid(*GRIDS["L14"])
    r = grid14.r()
    N = grid14.N
    base = _pots(grid14)["screened86"]
    rng = np.random.default_rng(7)
    pots = {
        "scaled_1e4": base * 1e4,
        "barrier": np.where((r > 0.5) & (r < 3.0), 5e3, base),
        "deep_well": np.where(r < 20.0, -400.0, 0.0),
        "nan_hole": np.where((np.arange(N) > 9000) & (np.arange(N) < 9004), np.nan, base),
        "inf_spike": np.where(np.arange(N) == 7000, np.inf, base),
        "zero": np.zeros(N),
    }
    names = list(pots)
    V = np.stack([pots[k] for k in names])
    nt = 64 * len(names) + 37
    vidx = (np.arange(nt) % len(names)).astype(np.int32)
    l = rng.integers(0, 4, nt).astype(np.int32)
    E = np.where(rng.random(nt) < 0.15, rng.uniform(0, 30, nt), -10.0 ** rng.uniform(-3, 3.5, nt))
    lim = rng.choice([0, 1, 3, 40, 100000], nt).astype(np.int32)
    c = D.numerov_sweeps_scan(ctx, grid14, D.SWEEP_COUNT, V, l, E, lim, vidx=vidx)
    decided, bad = 0, []
    for k in range(nt):
        st, tr = C.c_long(), C.c_long()
        want = o.dfo_count_nodes(C.byref(g), O.dp(V[vidx[k]]), int(l[k]), float(E[k]), int(lim[k]), C.byref(st), C.byref(tr))
        assert c["start"][k] == st.value
        if c["fallback"][k]:
            continue
        decided += 1
        if c["count"][k] != want:
            bad.append((names[vidx[k]], int(l[k]), float(E[k]), int(lim[k]), int(c["count"][k]), want))
    print("pathological potentials: %d of %d trials decided by the scan, %d differ" % (decided, nt, len(bad)))
    assert decided >= nt // 3
    assert not bad, bad[:5]


def test_scan_sweeps_full_size_equal_exact_kernels(ctx, grid17):
    """131 073 nodes: 2048 trials across the spectrum of a screened Rn potential -- counts, cut-offs equal the exact kernels', u(0) 1e-6
    relative at worst (the rows next to a zero of u(0)), 1e-10 in the median"""
    V = screened_potential(grid17.r(), 86.0)
    nt = 2048
    E = -10.0 ** np.linspace(-1, 3.5, nt)
    l = (np.arange(nt) % 4).astype(np.int32)
    lim = np.full(nt, 3, np.int32)
    a = D.numerov_sweeps(ctx, grid17, D.SWEEP_COUNT, V, l, E, lim, boundary=D.BOUNDARY_DEVICE)
    b = D.numerov_sweeps_scan(ctx, grid17, D.SWEEP_COUNT, V, l, E, lim)
    assert np.array_equal(a["count"], b["count"]) and np.array_equal(a["start"], b["start"]) and not b["fallback"].any()
    za = D.numerov_sweeps(ctx, grid17, D.SWEEP_ZERO, V, l, E, boundary=D.BOUNDARY_DEVICE)
    zb = D.numerov_sweeps_scan(ctx, grid17, D.SWEEP_ZERO, V, l, E)
    rel = np.abs(za["u0"] - zb["u0"]) / np.abs(za["u0"])
    assert np.nanmax(rel) <= 1e-6 and np.nanmedian(rel) <= 1e-10, (np.nanmax(rel), np.nanmedian(rel))


def _rn_levels():
    return D.get_subshells(86)


@pytest.mark.parametrize("mode", [D.LEVELS_BATCHED, D.LEVELS_CHAINED])
def test_scan_level_search_vs_exact(ctx, grid17, mode):
    """LoopOverLevels for the 15 levels of Rn on a screened potential, both bracket modes: eigenvalues within the round-off band of the
    reference's rounding bias of the exact path (6e-11 |E| + 6e-10 Ha; observed 3.4e-11 |E| for 1s, 2.5e-10 Ha for the outer levels), the
    same numbers of reference-equivalent sweeps within 2 %, the same convergence flags, densities 1e-8 relative"""
    V = screened_potential(grid17.r(), 86.0)
    lv = _rn_levels()
    a = D.solve_levels(ctx, grid17, V, lv, -86.0 ** 2 - 1, mode=mode)
    b = D.solve_levels(ctx, grid17, V, lv, -86.0 ** 2 - 1, mode=mode | D.LEVELS_SCAN_SWEEPS)
    dE = np.abs(a["E"] - b["E"])
    print("scan level search (mode %d): max |dE| / |E| = %.2e, max |dE| = %.2e Ha; sweeps %d / %d (count), %d / %d (zero)"
          % (mode, np.max(dE / np.abs(a["E"])), dE.max(), a["n_count"].sum(), b["n_count"].sum(), a["n_zero"].sum(), b["n_zero"].sum()))
    assert np.all(dE <= 6e-11 * np.abs(a["E"]) + 6e-10), dE
    assert abs(int(a["n_count"].sum()) - int(b["n_count"].sum())) <= 0.02 * a["n_count"].sum()
    assert abs(int(a["n_zero"].sum()) - int(b["n_zero"].sum())) <= 0.02 * a["n_zero"].sum() + 2
    assert np.array_equal(a["converged"], b["converged"])
    nd_a, nd_b = a["newDensity"][0], b["newDensity"][0]
    assert np.max(np.abs(nd_a - nd_b)) <= 1e-8 * np.max(np.abs(nd_a))


def _decisions(lo, hi, end, ends_on_hi):
    """the decision string of a bisection of DFTAtom.cpp:566-604 / 517-534 from its start interval and its end point: a step that
    moves BottomEnergy up to the midpoint m leaves the end point above m (at or above it, when the end point is BottomEnergy itself)"""
    bits, widths = [], []
    for _ in range(80):
        if not (hi - lo > 1e-12):
            break
        m = (hi + lo) / 2
        up = (end > m) if ends_on_hi else (end >= m)
        bits.append(bool(up))
        widths.append(hi - lo)
        if up:
            lo = m
        else:
            hi = m
    return bits, widths


def test_scan_and_exact_bisections_agree_outside_the_round_off_band(ctx, grid17):
    """north_star's "node counts bit-exact" holds for the exact kernels; for the scan sweeps it holds OUTSIDE a band around every count
    transition and every sign change of u(0) -- inside it the reference's own count is round-off (the count flips hundreds of
    times within 1.2e-12 .. 7.4e-12 |E|) and a bisection converges INTO that band by construction.  Here the statement is asserted on the
    energies the search really visits: the three bisections of all 15 levels of Rn run on the scan from the start intervals of the
    reference's chain; every midpoint of the scan's path (reconstructed from its end points with the reference's (toe + boe) / 2) is
    integrated AGAIN by the exact kernels, and the exact kernels' decision -- count > n, count < n, sign of u(0) against the sign at
    BottomEnergy -- must be the scan's wherever the midpoint is further than 6e-11 |T| + 6e-10 Ha from the transition T the exact
    search itself ends on: the gate of the eigenvalues of the two paths (test_scan_level_search_vs_exact; T is itself a point inside
    the exact kernels' band, and the scan's transition sits up to the rounding bias of the reference's recurrence away from it:
    tests/test_scan_precision.py).  The test prints how many of the ~2000 decisions differ and how far from T the furthest one is."""
    V = screened_potential(grid17.r(), 86.0)
    lv = _rn_levels()
    chained = D.solve_levels(ctx, grid17, V, lv, -86.0 ** 2 - 1, mode=D.LEVELS_CHAINED)
    hints = np.concatenate(([-86.0 ** 2 - 1], chained["E"][:-1] - 3.0))             # DFTAtom.cpp:407,541
    a = D.solve_levels(ctx, grid17, V, lv, -86.0 ** 2 - 1, mode=D.LEVELS_BATCHED, hints=hints)
    b = D.solve_levels(ctx, grid17, V, lv, -86.0 ** 2 - 1, mode=D.LEVELS_BATCHED | D.LEVELS_SCAN_SWEEPS, hints=hints)
    assert np.array_equal(a["E"].view(np.int64), chained["E"].view(np.int64))        # the hints reproduce the reference's chain
    Es, ls, lims, tags = [], [], [], []
    for k, (n, l, _) in enumerate(lv):
        nodes = n - l
        # (start interval, the scan's end point, the exact search's end point = the transition, which end the bisection returns)
        phases = ((hints[k], 50.0, b["top"][k], a["top"][k], True), (hints[k], b["top"][k], b["bottom"][k], a["bottom"][k], True),
                  (b["bottom"][k], b["top"][k], b["E"][k], a["E"][k], False))
        for ph, (lo, hi, end, T, on_hi) in enumerate(phases):
            if nodes == 0 and ph == 1:
                continue                                                             # "count < 0": arithmetic on both paths
            bits, _ = _decisions(lo, hi, end, on_hi)
            if ph == 2:                                                              # DFTAtom.cpp:513: the sign at BottomEnergy first
                Es.append(lo); ls.append(l); lims.append(nodes); tags.append((k, ph, -1, None, T))
            for i, bit in enumerate(bits):
                m = (hi + lo) / 2
                Es.append(m); ls.append(l); lims.append(nodes); tags.append((k, ph, i, bit, T))
                if bit:
                    lo = m
                else:
                    hi = m
    Es, ls, lims = np.array(Es), np.array(ls, np.int32), np.array(lims, np.int32)
    cnt = D.numerov_sweeps(ctx, grid17, D.SWEEP_COUNT, V, ls, Es, lims, boundary=D.BOUNDARY_DEVICE)["count"]
    u0 = D.numerov_sweeps(ctx, grid17, D.SWEEP_ZERO, V, ls, Es, boundary=D.BOUNDARY_DEVICE)["u0"]
    sgn_bottom, worst, ndiff, ntot = {}, 0.0, 0, 0
    for q, (k, ph, i, bit, T) in enumerate(tags):
        nodes = lims[q]
        if i < 0:
            sgn_bottom[k] = u0[q] > 0
            continue
        exact = (not (cnt[q] > nodes)) if ph == 0 else ((cnt[q] < nodes) if ph == 1 else ((u0[q] > 0) == sgn_bottom[k]))
        ntot += 1
        if bool(exact) != bool(bit):
            ndiff += 1
            dist = abs(Es[q] - T)
            worst = max(worst, dist / abs(T))
            assert dist <= 6e-11 * abs(T) + 6e-10, (lv[k], ph, i, Es[q], T, dist / abs(T))
    print("scan against exact decisions at the scan's own %d midpoints: %d differ, all within %.1e |T| of the transition" % (ntot, ndiff, worst))
    assert ntot > 1500


@pytest.mark.parametrize("lsda", [False, True])
def test_scan_mode_radon_steps_vs_reference(ctx, grid17, lsda):
    """BASELINE configs[1] / [2] with the sweeps in tolerance mode (multigrid exact) against the compiled reference's golden steps, and
    with BOTH tolerance modes along the whole recorded trajectory"""
    gold = json.load(open(os.path.join(HERE, "golden", "rn_end_to_end.json")))["Rn_LSDA_L17" if lsda else "Rn_LDA_L17"]
    scf = D.Scf(ctx, grid17, [86], lsda=lsda, levels_mode=D.LEVELS_CHAINED, sweep_mode=D.SWEEPS_TOLERANCE)
    worst_e, worst_l = 0.0, 0.0
    for key in ("first", "second"):
        st = scf.step()
        assert st.levels_layout == 4                      # the scan ran (no hand-back to the exact kernels)
        want = gold[key]
        en = scf.energies()[0][0].as_list()
        lv = np.concatenate([scf.levels(0, 0)["E"]] + ([scf.levels(0, 1)["E"]] if lsda else []))
        wl = np.array([x[1] for x in want["levels"]])
        de = max(abs(a - b) / abs(b) for a, b in zip(en, want["energies"]))
        dl = np.abs(lv - wl)
        worst_e = max(worst_e, de)
        worst_l = max(worst_l, float(np.max(dl / np.abs(wl))))
        assert de <= 1e-9, (key, de)
        assert np.all(dl <= 1e-8 + 2e-9 * np.abs(wl)), (key, dl.max())
    scf.close()
    traj = np.array(gold["etotal_all"])
    for pm in (D.POISSON_EXACT, D.POISSON_TOLERANCE, D.POISSON_ADAPTIVE):
        scf2 = D.Scf(ctx, grid17, [86], lsda=lsda, sweep_mode=D.SWEEPS_TOLERANCE, poisson_mode=pm)
        got, vcs = [], []
        for _ in range(len(traj)):
            st = scf2.step()
            if st.vcycles > 0:                      # (a step after the atom has met the stop test integrates nothing)
                vcs.append(int(st.vcycles))
            got.append(scf2.energies()[0][0].Etotal)
        rel = np.abs(np.array(got) - traj) / np.abs(traj)
        print("scan sweeps, Rn %s, multigrid %s: steps 0/1 energies %.2e rel, eigenvalues %.2e |E|; trajectory (%d steps) %.2e; V-cycles per solve %d .. %d"
              % ("LSDA" if lsda else "LDA", ("exact", "tolerance", "adaptive")[pm], worst_e, worst_l, len(traj), rel.max(), min(vcs), max(vcs)))
        assert rel.max() <= 2e-9
        # the reference's stop test lies below the cycle's round-off floor: it always runs to its cap; the adaptive mode stops on the floor
        assert (max(vcs) <= 12) if pm == D.POISSON_ADAPTIVE else (min(vcs) == 100)
        scf2.close()


def test_scan_mode_small_grid_and_batch(ctx):
    """4097 nodes (8 rows per lane: the per-lane row loops) and a batch of atoms: Ar, Ne, Kr to the end of their SCF against the exact path"""
    L, d, R = 12, 2e-3, 25.0
    grid = D.Grid(ctx, L, d, R)
    Z = [18, 10, 36]
    a = D.Scf(ctx, grid, Z)
    b = D.Scf(ctx, grid, Z, sweep_mode=D.SWEEPS_TOLERANCE)
    for k in range(12):
        a.step(want_stats=False)
        b.step(want_stats=False)
    ea, eb = a.energies()[0], b.energies()[0]
    for i in range(len(Z)):
        for x, y in zip(ea[i].as_list(), eb[i].as_list()):
            assert abs(x - y) <= 2e-9 * abs(x), (Z[i], x, y)
        la, lb = a.levels(i, 0)["E"], b.levels(i, 0)["E"]
        assert np.all(np.abs(la - lb) <= 1e-8 + 2e-9 * np.abs(la))
    a.close()
    b.close()
    grid.close()


def test_scan_groups_take_the_same_decisions(ctx, grid17):
    """k_scan_levels_group (15 / 7 / 3 workgroups per level, a depth-4 / 3 / 2 bisection tree per round) against one workgroup per level
    (SCAN_GROUP=1): the same reference decisions, so eigenvalues, intervals, sweep counts, status bits and the density bit for bit -- Rn
    LDA (15 levels x 15 members) and LSDA (30 levels x 7), six SCF steps each.  From the third step on a round starts with a spine of
    decisions predicted from the history bracket (SCAN_NOSPINE switches it off): a prediction only selects which midpoints are integrated
    together, so the run without spines is the same bit for bit as well"""
    def run(knob, lsda):
        old = os.environ.get("DFTA_DEBUG")
        try:
            if knob:
                os.environ["DFTA_DEBUG"] = knob
            else:
                os.environ.pop("DFTA_DEBUG", None)
            scf = D.Scf(ctx, grid17, [86], lsda=lsda, sweep_mode=D.SWEEPS_TOLERANCE)
            out = []
            for _ in range(6):
                st = scf.step()
                lv = [scf.levels(0, sp) for sp in range(2 if lsda else 1)]
                out.append((scf.energies()[0][0].as_list(), [x["E"].copy() for x in lv], [x["status"].copy() for x in lv], [x["n_count"].copy() for x in lv],
                            [x["n_zero"].copy() for x in lv], int(st.sweeps_reference), int(st.points_reference), scf.array(0).copy(), float(st.ms_levels)))
            scf.close()
            return out
        finally:
            if old is None:
                os.environ.pop("DFTA_DEBUG", None)
            else:
                os.environ["DFTA_DEBUG"] = old
    for lsda in (False, True):
        a = run(None, lsda)
        for knob in ("SCAN_GROUP=1", "SCAN_GROUP=3", "SCAN_NOSPINE", "SCAN_GROUP=7,SCAN_NOSPINE"):
            b = run(knob, lsda)
            for x, y in zip(a, b):
                assert x[0] == y[0] and x[5:7] == y[5:7] and np.array_equal(x[7], y[7]), knob
                for q in range(1, 5):
                    assert all(np.array_equal(u, v) for u, v in zip(x[q], y[q])), (knob, q)
        print("scan level search Rn %s: %.2f ms per step grouped, %.2f ms with one workgroup per level" % ("LSDA" if lsda else "LDA", a[-1][8], run("SCAN_GROUP=1", lsda)[-1][8]))
