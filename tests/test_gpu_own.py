"""GPU suite (-m gpu): the own-pace level search of a BATCH (dftatom_amd/csrc/own.inc: more than 64 live levels, one workgroup of W
waves per level in ONE ordinary launch -- plan, expand, fused sweeps, scouts, walk inside the kernel, no workgroup waits for another)
against the host-synchronised rounds of levels.hip, which run the same device functions (levels_device.inc, numerov.hip:sweep_wave)
in lock step -- DFTAtom.cpp:493-604, Numerov.h:272-401.

The bar is bit-identity, as for the device-side search of a single atom (test_gpu_persist.py): energies, eigenvalues, convergence flags
and status bits, the reference-equivalent sweep counts, the traversed points of the sweeps on the reference's path, densities and
potentials of every SCF step.  Which midpoints are integrated speculatively differs; no decision does.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import dftatom_amd as D                 # noqa: E402
from golden.make_golden import GRIDS    # noqa: E402
from test_gpu_persist import _Knobs, _assert_same, _run    # noqa: E402


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


def test_own_pace_search_equals_host_rounds_batch(ctx, grid14):
    """Z = 1..24 (98 levels, l = 0..2) at 16 385 nodes, eight steps: layout 6 against the host rounds (static blocks)"""
    Z = list(range(1, 25))
    dev = _run(ctx, grid14, Z, False, 8, "LEVELS_OWN")
    host = _run(ctx, grid14, Z, False, 8, "LEVELS_NOPERSIST")
    assert all(r["layout"] == 6 for r in dev), [r["layout"] for r in dev]
    assert all(r["layout"] in (0, 2) for r in host), [r["layout"] for r in host]
    _assert_same(dev, host, "Z=1..24")


def test_own_pace_search_lsda_open_shells(ctx, grid14):
    """spin-polarised batch with open shells, d and f levels (Fe, Gd, Cu, N, Ar): 2 x the levels, slots per spin"""
    Z = [26, 64, 29, 7, 18, 36]
    dev = _run(ctx, grid14, Z, True, 6, "LEVELS_OWN")
    host = _run(ctx, grid14, Z, True, 6, "LEVELS_NOPERSIST")
    assert all(r["layout"] == 6 for r in dev), [r["layout"] for r in dev]
    _assert_same(dev, host, "LSDA batch")


@pytest.mark.parametrize("knobs", ["LEVELS_OWN_WMAX=1", "LEVELS_OWN_WMAX=2", "LEVELS_OWN_WAVES=256", "LEVELS_OWN_SPINE_CAP=48,LEVELS_OWN_WMAX=1",
                                   "LEVELS_OWN_SPINE_CAP=8", "LEVELS_NOPREDICT", "LEVELS_NOEXTRAP"])
def test_layout_knobs_of_the_own_pace_search_keep_the_bits(ctx, grid14, knobs):
    """one wave per level, two, an eighth of the machine, long and short spines, no predictions at all: rounds change, results do not"""
    Z = [36, 30, 18, 10, 54, 12, 20, 38, 47, 5]
    ref = _run(ctx, grid14, Z, False, 5, "LEVELS_NOPERSIST")
    alt = _run(ctx, grid14, Z, False, 5, "LEVELS_OWN," + knobs)
    assert all(r["layout"] == 6 for r in alt), [r["layout"] for r in alt]
    _assert_same(ref, alt, knobs)


def test_batch_to_its_end_through_all_three_searches(ctx, grid14):
    """a batch of 20 atoms (74 levels) starts in the own-pace search (layout 6); once at most 64 levels are live the single-atom device
    search takes over (layout 5); every atom ends in the state it reaches alone with host rounds"""
    Z = list(range(1, 21))
    with _Knobs("LEVELS_OWN"):
        batch = D.Scf(ctx, grid14, Z, lsda=False)
    layouts = []
    for _ in range(110):
        st = batch.step()
        layouts.append(int(st.levels_layout))
        _, fin = batch.energies()
        if fin.all():
            break
    eb, _ = batch.energies()
    assert layouts[0] == 6 and 5 in layouts, sorted(set(layouts))
    for ia in (0, 5, 10, 19):
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


def test_periodic_table_two_steps_full_size(ctx, grid17):
    """BASELINE configs[3] on one GPU: Z = 1..86 at 131 073 nodes (738 levels, W = 2), two SCF steps against the packed host rounds"""
    Z = list(range(1, 87))
    dev = _run(ctx, grid17, Z, False, 2, "LEVELS_OWN")
    host = _run(ctx, grid17, Z, False, 2, "LEVELS_NOPERSIST")
    assert all(r["layout"] == 6 for r in dev), [r["layout"] for r in dev]
    assert all(r["layout"] == 2 for r in host), [r["layout"] for r in host]
    _assert_same(dev, host, "Z=1..86 @ 131 073")


def test_rn_lsda_batch_at_2_20_nodes(ctx):
    """BASELINE configs[4]: 8 x Rn LSDA at 1 048 577 nodes (240 levels, W = 8), two steps"""
    L, d, R = GRIDS["L20"]
    g = D.Grid(ctx, L, d, R)
    try:
        dev = _run(ctx, g, [86] * 8, True, 2, "LEVELS_OWN")
        host = _run(ctx, g, [86] * 8, True, 2, "LEVELS_NOPERSIST")
    finally:
        g.close()
    assert all(r["layout"] == 6 for r in dev), [r["layout"] for r in dev]
    _assert_same(dev, host, "8 x Rn LSDA @ 2^20+1")


def test_longest_first_launch_on_static_blocks_keeps_the_bits(ctx, grid14):
    """the default batch path: Z = 75..86 (170 levels, two 64-trial blocks each: 340 blocks on 256 compute units, the pipelined kernel) with
    the blocks of a round launched longest first (k_expand's queue) against the plain launch in array order (LEVELS_NOQUEUE): same bits --
    the order in which blocks start is invisible in every result"""
    Z = list(range(75, 87))
    # (LEVELS_PERSIST_WIDE=64: round 5's limit of the device-side search, which takes batches of up to 256 levels by default)
    q = _run(ctx, grid14, Z, False, 6, "LEVELS_PERSIST_WIDE=64")
    plain = _run(ctx, grid14, Z, False, 6, "LEVELS_PERSIST_WIDE=64,LEVELS_NOQUEUE")
    assert all(r["layout"] == 0 for r in q) and all(r["layout"] == 0 for r in plain), ([r["layout"] for r in q], [r["layout"] for r in plain])
    _assert_same(q, plain, "queued vs plain launch")
    # ... and the scan predictor of the first spines (one workgroup per level ahead of the rounds): off / shifted -- fewer or more rounds, same bits
    nopred = _run(ctx, grid14, Z, False, 6, "LEVELS_PERSIST_WIDE=64,LEVELS_NOSCANPREDICT_BATCH")
    wrong = _run(ctx, grid14, Z, False, 6, "LEVELS_PERSIST_WIDE=64,LEVELS_SCAN_PREDICT_SHIFT=-1e-4")
    dev = _run(ctx, grid14, Z, False, 6, "")                              # ... and the default: the device-side search of 174 levels
    assert all(r["layout"] == 5 for r in dev), [r["layout"] for r in dev]
    _assert_same(q, dev, "host rounds vs the device-side search")
    _assert_same(q, nopred, "scan predictor off")
    _assert_same(q, wrong, "scan predictor shifted")
    assert sum(r["rounds"] for r in q) < sum(r["rounds"] for r in nopred)
