"""GPU suite (-m gpu): the exact level search ON THE DEVICE (dftatom_amd/csrc/persist.inc: one persistent kernel, every level at its own
pace, the closing workgroup of a level walks, plans, matches and normalises) against the host-synchronised rounds of levels.hip, which
run the same device functions (levels_device.inc) in lock step -- DFTAtom.cpp:493-604, Numerov.h:272-504, DFTAtom.cpp:36-56.

The bar is bit-identity: energies, eigenvalues, convergence flags and status bits, the reference-equivalent sweep counts, the traversed
points of the sweeps on the reference's path, densities and wavefunction-derived quantities of every SCF step.  Which midpoints are
integrated speculatively differs between the two (a level plans from what its sibling has reached at that moment); no decision does.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import dftatom_amd as D                 # noqa: E402
from golden.make_golden import GRIDS    # noqa: E402


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


class _Knobs:
    """DFTA_DEBUG for the duration of a block (the knobs are read when a solver is created)"""

    def __init__(self, value):
        self.value = value

    def __enter__(self):
        self.old = os.environ.get("DFTA_DEBUG")
        if self.value:
            os.environ["DFTA_DEBUG"] = self.value
        else:
            os.environ.pop("DFTA_DEBUG", None)

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("DFTA_DEBUG", None)
        else:
            os.environ["DFTA_DEBUG"] = self.old


def _run(ctx, grid, Z, lsda, nsteps, knobs):
    with _Knobs(knobs):
        scf = D.Scf(ctx, grid, Z, lsda=lsda)
    out = []
    for _ in range(nsteps):
        st = scf.step()
        en, fin = scf.energies()
        rec = {"layout": int(st.levels_layout), "rounds": int(st.rounds), "sweeps_reference": int(st.sweeps_reference),
               "sweeps_reference_executed": int(st.sweeps_reference_executed), "points_reference": int(st.points_reference), "atoms": []}
        for a in range(len(Z)):
            lv = [scf.levels(a, s) for s in range(2 if lsda else 1)]
            rec["atoms"].append({"E": en[a].as_list(), "fin": int(fin[a]), "levels": lv, "rho": scf.array(0, a), "U": scf.array(5, a)})
        out.append(rec)
    scf.close()
    return out


def _assert_same(a, b, what):
    assert len(a) == len(b)
    for k, (x, y) in enumerate(zip(a, b)):
        for key in ("sweeps_reference", "sweeps_reference_executed", "points_reference"):
            assert x[key] == y[key], (what, k, key, x[key], y[key])
        for ia, (p, q) in enumerate(zip(x["atoms"], y["atoms"])):
            assert p["E"] == q["E"], (what, k, ia)
            assert p["fin"] == q["fin"], (what, k, ia)
            for lp, lq in zip(p["levels"], q["levels"]):
                assert np.array_equal(lp["E"].view(np.int64), lq["E"].view(np.int64)), (what, k, ia, lp["E"] - lq["E"])
                for key in ("converged", "status", "n_count", "n_zero"):
                    assert np.array_equal(lp[key], lq[key]), (what, k, ia, key)
            assert np.array_equal(p["rho"].view(np.int64), q["rho"].view(np.int64)), (what, k, ia)
            assert np.array_equal(p["U"].view(np.int64), q["U"].view(np.int64)), (what, k, ia)


def test_device_search_equals_host_rounds_radon(ctx, grid17):
    """BASELINE configs[1]: Rn LDA at 131 073 nodes, eight SCF steps -- the default path (layout 5) against LEVELS_NOPERSIST (layout 1)"""
    dev = _run(ctx, grid17, [86], False, 8, "")
    host = _run(ctx, grid17, [86], False, 8, "LEVELS_NOPERSIST")
    assert all(r["layout"] == 5 for r in dev), [r["layout"] for r in dev]
    assert all(r["layout"] == 1 for r in host), [r["layout"] for r in host]
    _assert_same(dev, host, "Rn LDA")


def test_device_search_equals_host_rounds_lsda(ctx, grid17):
    """BASELINE configs[2]: Rn LSDA (30 levels: eight workgroups each)"""
    dev = _run(ctx, grid17, [86], True, 4, "")
    host = _run(ctx, grid17, [86], True, 4, "LEVELS_NOPERSIST")
    assert all(r["layout"] == 5 for r in dev)
    _assert_same(dev, host, "Rn LSDA")


@pytest.mark.parametrize("Z,lsda", [([1], False), ([2], False), ([18], True), ([26], False), ([64], False)])
def test_device_search_equals_host_rounds_small_grid(ctx, grid14, Z, lsda):
    """one level (H), one doubly occupied level (He), open shells, d and f levels at 16 385 nodes: to the stop or twelve steps"""
    dev = _run(ctx, grid14, Z, lsda, 12, "")
    host = _run(ctx, grid14, Z, lsda, 12, "LEVELS_NOPERSIST")
    assert dev[0]["layout"] == 5
    _assert_same(dev, host, "Z=%d" % Z[0])


@pytest.mark.parametrize("knobs", ["LEVELS_PERSIST_NOCAND", "LEVELS_PERSIST_EQUAL", "LEVELS_PERSIST_BLOCKS=64", "LEVELS_PERSIST_BLOCKS=37,LEVELS_PERSIST_EQUAL",
                                   "LEVELS_NOPREDICT", "LEVELS_PERSIST_PLAIN_LAUNCH", "LEVELS_PERSIST_NOBUDGET", "LEVELS_NOSCANPREDICT",
                                   "LEVELS_SCAN_PREDICT_SHIFT=1e-4"])
def test_layout_knobs_of_the_device_search_keep_the_bits(ctx, grid14, knobs):
    """no speculative match solves, equal shares, a quarter of the machine, an odd number of workgroups, no predictions at all, an
    ordinary launch, no candidate budget, no scan predictor of the first spines, a wrong one: rounds change, results do not"""
    ref = _run(ctx, grid14, [36], False, 5, "")
    alt = _run(ctx, grid14, [36], False, 5, knobs)
    assert all(r["layout"] == 5 for r in alt)
    _assert_same(ref, alt, knobs)


def test_batch_whose_last_atoms_search_on_the_device(ctx, grid14):
    """a batch of 45 atoms (280 levels) starts in packed host rounds; once at most 256 levels are live its rounds move to the device (layout 5:
    one or two workgroups per level, later the floating shares of a single atom's search), and every atom ends in the state it reaches
    alone with host rounds"""
    Z = list(range(1, 46))
    with _Knobs(""):
        batch = D.Scf(ctx, grid14, Z, lsda=False)
    layouts = []
    for _ in range(110):
        st = batch.step()
        layouts.append(int(st.levels_layout))
        _, fin = batch.energies()
        if fin.all():
            break
    eb, _ = batch.energies()
    assert 5 in layouts and layouts[0] != 5, sorted(set(layouts))
    for ia in (0, 7, 12, 19, 28, 44):
        with _Knobs("LEVELS_NOPERSIST"):
            one = D.Scf(ctx, grid14, [Z[ia]], lsda=False)
        for _ in range(110):
            one.step()
            e1, f1 = one.energies()
            if f1[0]:
                break
        assert e1[0].as_list() == eb[ia].as_list(), Z[ia]
        lb, l1 = batch.levels(ia, 0), one.levels(0, 0)
        assert np.array_equal(lb["E"].view(np.int64), l1["E"].view(np.int64)), Z[ia]
        one.close()
    batch.close()


def test_device_search_of_a_batch_with_65_to_128_live_levels(ctx, grid14, grid17):
    """Round 6: the device-side search takes up to 256 live levels -- one workgroup per level, a second one for as many levels as there are
    compute units left (all of them up to 128 levels: the shards of an 8-rank periodic table start with ~100), every level at its own
    pace, match solve and normalisation inside.  Eleven atoms (76 / 105 levels) at 16 385 and at 131 073 nodes and 21 atoms (145 levels: 111
    second workgroups, handed to the levels that ended last in the previous steps) against the host rounds (LEVELS_NOPERSIST) and against
    round 5's limit (LEVELS_PERSIST_WIDE=64: static blocks); a lost worker sends the step back to the batch's own host rounds (layout 0)."""
    Z14 = list(range(20, 31))                                            # 76 levels
    wide14 = _run(ctx, grid14, Z14, False, 6, "")
    host14 = _run(ctx, grid14, Z14, False, 6, "LEVELS_NOPERSIST")
    narrow14 = _run(ctx, grid14, Z14, False, 6, "LEVELS_PERSIST_WIDE=64")
    assert all(r["layout"] == 5 for r in wide14), [r["layout"] for r in wide14]
    assert all(r["layout"] == 0 for r in host14) and all(r["layout"] == 0 for r in narrow14)
    _assert_same(wide14, host14, "wide vs host rounds")
    _assert_same(wide14, narrow14, "wide vs the 64-level limit")
    Z17 = [3, 11, 19, 30, 37, 48, 55, 62, 70, 79, 86]                    # 105 levels
    wide17 = _run(ctx, grid17, Z17, False, 3, "")
    host17 = _run(ctx, grid17, Z17, False, 3, "LEVELS_NOPERSIST")
    assert all(r["layout"] == 5 for r in wide17) and all(r["layout"] == 0 for r in host17)
    _assert_same(wide17, host17, "wide vs host rounds @ 131 073")
    Z21 = list(range(10, 31))                                            # 145 levels: one workgroup each + 111 second ones
    wide21 = _run(ctx, grid14, Z21, False, 5, "")
    host21 = _run(ctx, grid14, Z21, False, 5, "LEVELS_NOPERSIST")
    mid21 = _run(ctx, grid14, Z21, False, 5, "LEVELS_PERSIST_EQUAL")     # one workgroup per level, the rest in the pool
    assert all(r["layout"] == 5 for r in wide21) and all(r["layout"] == 5 for r in mid21) and all(r["layout"] == 0 for r in host21)
    _assert_same(wide21, host21, "145 levels vs host rounds")
    _assert_same(mid21, host21, "145 levels, equal shares, vs host rounds")
    lost = _run(ctx, grid14, Z14, False, 3, "FAULT_PERSIST_WORKER=1,LEVELS_PERSIST_TIMEOUT_MS=300")
    assert lost[0]["layout"] == 0, [r["layout"] for r in lost]
    _assert_same(lost, host14[:3], "lost worker of a wide search")


def test_lost_worker_is_detected_and_the_solve_repeated(ctx, grid14):
    """FAULT_PERSIST_WORKER: one workgroup never answers its first message; the closer's arrival count stays short, the bounded waits
    raise the abort flag, the host repeats the solve with host rounds -- same bits, layout 1 for that step"""
    ref = _run(ctx, grid14, [18], False, 3, "LEVELS_NOPERSIST")
    alt = _run(ctx, grid14, [18], False, 3, "FAULT_PERSIST_WORKER=1,LEVELS_PERSIST_TIMEOUT_MS=300")
    assert alt[0]["layout"] == 1 and alt[1]["layout"] == 1, [r["layout"] for r in alt]
    _assert_same(ref, alt, "lost worker")
