"""GPU suite (-m gpu): the resident multigrid groups (k_poisson_solve_res: 32 member workgroups keep their stretch of every
shared level in LDS, one fused three-sweep pass and one exchange per visit, a coarse workgroup below; for 8 .. 15 atoms: 16 members
whose level 0 takes turns with their other shared levels) and the opt-in tolerance mode of the smoother.

EXACT mode (default): U, the V-cycle count and the last error norm are the bits of the one-workgroup solve -- which equals the
reference's PoissonSolver (tests/test_gpu_parity.py, test_oracle_vs_ref.py) -- for every grid the resident layout serves
(16385 .. 131073 nodes: 1 .. 4 shared levels), for Z = 1 (the cycle stops early: the visits' stop-after-one-sweep path) to 86,
for batches of 1 .. 7 atoms (33 workgroups each on 256 compute units), run after run (the exchange slots are validated by content, never by timing), and through a whole
SCF.  TOLERANCE mode: U within 2e-9 Z of the exact solve (observed 9.1e-10 Z at Z = 1, 3.5e-10 Z at Z = 86: the end state of the cycle is a round-off floor that
any perturbation of the iteration shifts by that much -- the reference moves by as much under FMA contraction,
test_oracle_golden.py::test_reference_rounding_sensitivity_of_scf_steps), SCF energies of the first steps within 1e-9 relative
(observed 1.1e-10) and eigenvalues within 1e-8 Ha + 2e-9 |E| (observed 3.0e-10 |E|) of the reference's golden values; Etotal along
the whole recorded trajectory within 2e-9 (observed 4.8e-10 LDA / 9.6e-10 LSDA; the exact mode shows 6.7e-10 of the same late-step
jitter, test_gpu_configs.py).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import dftatom_amd as D                 # noqa: E402
from golden.make_golden import GRIDS    # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ctx(torch_first):
    c = D.Context(0)
    yield c
    c.close()


from _knobs import knobs as env          # noqa: E402  (DFTA_DEBUG entries for the duration of a block)


def _solve(ctx, grid, Zs, rho, mode=D.POISSON_DEFAULT, **kv):
    with env(**kv):
        ps = D.Poisson(ctx, grid, len(Zs), mode)
        U, vc, err = ps.solve(Zs, rho)
        info = ps.group_info()
        ps.close()
    return U, vc, err, info


@pytest.mark.parametrize("L,delta,R", [(14, 5e-4, 25.0), (15, 2.5e-4, 30.0), (16, 2e-4, 40.0), (17, 1e-4, 50.0)])
def test_resident_groups_return_the_one_workgroup_bits(ctx, L, delta, R):
    grid = D.Grid(ctx, L, delta, R)
    rr = grid.r()
    for Zs in ([86], [1], [18, 2], [2, 54, 86, 7], [86, 36, 18, 10, 2, 1, 54]):
        rho = np.stack([z * (1.0 + 0.3 * k) ** 3 * np.exp(-2 * (1.0 + 0.3 * k) * rr) / np.pi for k, z in enumerate(Zs)])
        U1, vc1, e1, i1 = _solve(ctx, grid, Zs, rho, DFTA_POISSON_GROUP="0", DFTA_POISSON_NOFUSE3="1")
        Ur, vcr, er, ir = _solve(ctx, grid, Zs, rho)
        assert i1[0] == 1 and ir == (33, False, 0), (i1, ir)          # one workgroup per atom vs 32 members + coarse workgroup
        assert np.array_equal(Ur.view(np.int64), U1.view(np.int64)), (L, Zs)
        # (the last error norm is a sum of squares taken in another order: it may differ in its last bit, never in a decision)
        assert np.array_equal(vcr, vc1) and np.all(np.abs(er - e1) <= 1e-14 * np.abs(e1)), (L, Zs, vcr, vc1, er, e1)
    grid.close()


@pytest.mark.parametrize("L,delta,R", [(12, 2e-3, 25.0), (14, 5e-4, 25.0), (17, 1e-4, 50.0), (17, None, 50.0)])
def test_register_coarse_levels_return_the_lds_bits(ctx, L, delta, R):
    """Round 5: in exact mode the six coarsest levels of a V-cycle (65 ... 3 nodes) keep Phi and S in registers through the whole visit
    sequence -- lane-shift sweeps, restriction and prolongation as lane arithmetic (poisson_kernels.inc: xw_section) -- instead of
    level-by-level sweeps on LDS copies (POISSON_NOXW).  The same operations on the same values in the same order: U, the V-cycle count
    and the reported error norm are the same bits, for the resident groups, a staged group and one workgroup per atom, for small and
    large Z (Z = 1 meets the reference's stop test early: the sweep counts of the early-stop rule are exercised), on a uniform grid too."""
    grid = D.Grid(ctx, L, delta, R)
    rr = grid.r()
    for Zs in ([86], [1], [18, 2, 54]):
        rho = np.stack([z * (1.0 + 0.3 * k) ** 3 * np.exp(-2 * (1.0 + 0.3 * k) * rr) / np.pi for k, z in enumerate(Zs)])
        for kv in ({}, {"DFTA_POISSON_GROUP": "0"}, {"DFTA_POISSON_RES": "0"}):
            Ux, vcx, ex, _ = _solve(ctx, grid, Zs, rho, D.POISSON_EXACT, **kv)
            Ul, vcl, el, _ = _solve(ctx, grid, Zs, rho, D.POISSON_EXACT, DFTA_POISSON_NOXW="1", **kv)
            assert np.array_equal(Ux.view(np.int64), Ul.view(np.int64)), (L, Zs, kv)
            assert np.array_equal(vcx, vcl) and np.array_equal(ex.view(np.int64), el.view(np.int64)), (L, Zs, kv, vcx, vcl)
    grid.close()


@pytest.mark.parametrize("L,delta,R", [(12, 2e-3, 25.0), (14, 5e-4, 25.0), (17, 1e-4, 50.0), (17, None, 50.0), (20, 1.25e-5, 50.0)])
def test_fused_visits_of_the_one_wave_levels_return_the_single_sweep_bits(ctx, L, delta, R):
    """Round 6: the three sweeps of a visit of the coarse section's 257 ... 1025-node levels run as ONE fused pass on the level's LDS copy
    (poisson_kernels.inc: cs_visit3 -> gs_lds3 with the 64 lanes of the wave: 112 + C + 2 dependent steps, one prologue, one norm, one
    wave fence instead of three of each) -- against POISSON_NOFUSE3_WAVE, the sweep-by-sweep visits.  Same arithmetic per node: U, the
    V-cycle count, the sweep-dependent error norm of level 0 are the same bits, for the resident groups, a staged group and one workgroup
    per atom; Z = 1 meets the reference's stop rule early on the coarse levels (the fall-back to single sweeps is exercised)."""
    grid = D.Grid(ctx, L, delta, R)
    rr = grid.r()
    for Zs in ([86], [1], [18, 2, 54]):
        rho = np.stack([z * (1.0 + 0.3 * k) ** 3 * np.exp(-2 * (1.0 + 0.3 * k) * rr) / np.pi for k, z in enumerate(Zs)])
        for kv in ({}, {"DFTA_POISSON_GROUP": "0"}, {"DFTA_POISSON_RES": "0"}):
            Uf, vcf, ef, _ = _solve(ctx, grid, Zs, rho, D.POISSON_EXACT, **kv)
            Us, vcs, es, _ = _solve(ctx, grid, Zs, rho, D.POISSON_EXACT, DFTA_POISSON_NOFUSE3_WAVE="1", **kv)
            assert np.array_equal(Uf.view(np.int64), Us.view(np.int64)), (L, Zs, kv)
            assert np.array_equal(vcf, vcs) and np.array_equal(ef.view(np.int64), es.view(np.int64)), (L, Zs, kv, vcf, vcs)
    grid.close()


@pytest.mark.parametrize("L,delta,R", [(14, 5e-4, 25.0), (15, 2.5e-4, 30.0), (16, 2e-4, 40.0), (17, 1e-4, 50.0), (17, None, 50.0)])
def test_second_resident_configuration_returns_the_one_workgroup_bits(ctx, L, delta, R):
    """Round 6: 8 .. 15 atoms per launch -- 16 members of 256 lanes + the coarse workgroup per atom (17 workgroups; namespace mg_exact16 of
    poisson.hip), the same lanes, nodes per lane, levels and passes as the 33-workgroup groups; level 0 and the other shared levels take
    turns in a member's LDS (level 0's Phi waits in the member's global scratch while the cycle is below it).  U, the V-cycle counts and
    the error norms against one workgroup per atom and against the staged groups of 16 (POISSON_RES16=0), for every grid the layout
    serves (1 .. 4 shared levels; a uniform grid too), Z = 1 and 2 (the visits' early-stop path) among the atoms."""
    grid = D.Grid(ctx, L, delta, R)
    rr = grid.r()
    for Zs in ([86, 1, 54, 2, 36, 18, 10, 7], [86, 80, 71, 64, 57, 47, 36, 30, 26, 18, 10, 6, 3, 2, 1]):
        rho = np.stack([z * (1.0 + 0.1 * k) ** 3 * np.exp(-2 * (1.0 + 0.1 * k) * rr) / np.pi for k, z in enumerate(Zs)])
        U1, vc1, e1, i1 = _solve(ctx, grid, Zs, rho, DFTA_POISSON_GROUP="0", DFTA_POISSON_NOFUSE3="1")
        Us, vcs, es, i_s = _solve(ctx, grid, Zs, rho, DFTA_POISSON_RES16="0")
        Ur, vcr, er, ir = _solve(ctx, grid, Zs, rho)
        assert i1[0] == 1 and i_s[0] == 16 and ir == (17, False, 0), (i1, i_s, ir)
        assert np.array_equal(Ur.view(np.int64), U1.view(np.int64)), (L, Zs)
        assert np.array_equal(Ur.view(np.int64), Us.view(np.int64)), (L, Zs)
        assert np.array_equal(vcr, vc1) and np.all(np.abs(er - e1) <= 1e-14 * np.abs(e1)), (L, Zs, vcr, vc1, er, e1)
    # forced for a small batch: against the 33-workgroup groups
    Zs = [86, 1]
    rho = np.stack([z * np.exp(-2 * rr) / np.pi for z in Zs])
    Ua, vca, ea, ia = _solve(ctx, grid, Zs, rho)
    Ub, vcb, eb, ib = _solve(ctx, grid, Zs, rho, DFTA_POISSON_RES16="1")
    assert ia == (33, False, 0) and ib == (17, False, 0), (ia, ib)
    assert np.array_equal(Ua.view(np.int64), Ub.view(np.int64)) and np.array_equal(vca, vcb)
    grid.close()


def test_second_resident_configuration_in_an_scf_batch_and_with_a_lost_member(ctx):
    """nine atoms at 32 769 nodes (two shared levels: the overlay at work), six SCF steps: U of every atom and step with the 17-workgroup
    groups (default) and with the staged groups (POISSON_RES16=0); twice (run-to-run identical).  Then a member that never arrives:
    detected, the solve repeated with one workgroup per atom, same bits."""
    grid = D.Grid(ctx, 15, 2.5e-4, 30.0)
    Zs = [36, 30, 18, 10, 54, 12, 20, 2, 47]
    runs = []
    for kv in ({}, {}, {"DFTA_POISSON_RES16": "0"}):
        with env(**kv):
            scf = D.Scf(ctx, grid, Zs, lsda=False)
            tr = []
            for _ in range(6):
                scf.step(want_stats=False)
                tr.append([scf.array(5, a).copy() for a in range(len(Zs))])
            runs.append((scf.poisson_info()[0], tr))
            scf.close()
    assert runs[0][0] == 17 and runs[2][0] == 16, (runs[0][0], runs[2][0])
    for k in range(6):
        for a in range(len(Zs)):
            for other in (1, 2):
                assert np.array_equal(runs[0][1][k][a].view(np.int64), runs[other][1][k][a].view(np.int64)), (k, a, other)
    grid.close()
    L, d, R = GRIDS["L17"]
    grid = D.Grid(ctx, L, d, R)
    rr = grid.r()
    Zs = [86, 54, 36, 18, 10, 2, 30, 47, 80]
    rho = np.stack([z * np.exp(-2 * rr) / np.pi for z in Zs])
    U0, vc0, _, info0 = _solve(ctx, grid, Zs, rho)
    assert info0 == (17, False, 0)
    with env(DFTA_FAULT_POISSON_MEMBER="1"):
        ps = D.Poisson(ctx, grid, len(Zs))
        U1, vc1, _ = ps.solve(Zs, rho)
        assert ps.group_info() == (17, True, 1)
        ps.close()
    assert np.array_equal(U0.view(np.int64), U1.view(np.int64)) and np.array_equal(vc0, vc1)
    grid.close()


def test_second_resident_configuration_in_tolerance_mode(ctx):
    """tolerance mode with 17 workgroups per atom (mg_tol16): the same lanes and chunks as the 33-workgroup groups, hence the bits of every
    atom solved alone there; within the mode's gate (2e-9 Z) of the exact solve; faster than the staged groups it replaces is bench business"""
    L, d, R = GRIDS["L17"]
    grid = D.Grid(ctx, L, d, R)
    rr = grid.r()
    Zs = [86, 54, 36, 18, 10, 2, 30, 47, 80, 1]
    rho = np.stack([z * (1.0 + 0.1 * k) ** 3 * np.exp(-2 * (1.0 + 0.1 * k) * rr) / np.pi for k, z in enumerate(Zs)])
    Ux, _, _, ix = _solve(ctx, grid, Zs, rho, D.POISSON_EXACT)
    Ut, vct, et, it = _solve(ctx, grid, Zs, rho, D.POISSON_TOLERANCE)
    Us, vcs, es, i_s = _solve(ctx, grid, Zs, rho, D.POISSON_TOLERANCE, DFTA_POISSON_RES16="0")
    assert ix == (17, False, 0) and it == (17, False, 0) and i_s[0] == 16, (ix, it, i_s)
    for k, z in enumerate(Zs):
        assert np.max(np.abs(Ut[k] - Ux[k])) <= 2e-9 * z, (z, np.max(np.abs(Ut[k] - Ux[k])))
        assert np.max(np.abs(Us[k] - Ux[k])) <= 2e-9 * z, (z, np.max(np.abs(Us[k] - Ux[k])))
        U1, vc1, e1, i1 = _solve(ctx, grid, [z], rho[k:k + 1], D.POISSON_TOLERANCE)
        assert i1 == (33, False, 0)
        assert np.array_equal(U1[0].view(np.int64), Ut[k].view(np.int64)) and vc1[0] == vct[k], (z, np.max(np.abs(U1[0] - Ut[k])), vc1, vct[k])
    Ua, vca, _, ia = _solve(ctx, grid, Zs, rho, D.POISSON_ADAPTIVE)
    assert ia == (17, False, 0)
    for k, z in enumerate(Zs):
        assert np.max(np.abs(Ua[k] - Ux[k])) <= 2e-8 * z and vca[k] <= 40, (z, np.max(np.abs(Ua[k] - Ux[k])), vca[k])
    grid.close()


@pytest.mark.parametrize("L,delta,R", [(12, 2e-3, 25.0), (14, 5e-4, 25.0), (17, 1e-4, 50.0), (17, None, 50.0), (20, 1.25e-5, 50.0)])
def test_restriction_from_the_staging_memory_returns_the_global_bits(ctx, L, delta, R):
    """Round 6: where two consecutive levels of one workgroup are staged (8 193 -> 4 097 -> 2 049 nodes), the restriction folded into the
    coarser level's copy-in reads the finer level from the staging memory -- where its visit has just left it -- instead of from its global
    copy (POISSON_NOFOLD_LDS: the global reads of round 5).  Same values, same arithmetic: U, V-cycle counts, error norms, for the resident
    groups (both configurations), a staged group and one workgroup per atom; Z = 1 stops early (the visit is staged a second time: from
    global memory, the staging memory having been overwritten)."""
    grid = D.Grid(ctx, L, delta, R)
    rr = grid.r()
    for Zs in ([86], [1], [18, 2, 54]):
        rho = np.stack([z * (1.0 + 0.3 * k) ** 3 * np.exp(-2 * (1.0 + 0.3 * k) * rr) / np.pi for k, z in enumerate(Zs)])
        for kv in ({}, {"DFTA_POISSON_GROUP": "0"}, {"DFTA_POISSON_RES": "0"}, {"DFTA_POISSON_RES16": "1"}):
            Uf, vcf, ef, _ = _solve(ctx, grid, Zs, rho, D.POISSON_EXACT, **kv)
            Us, vcs, es, _ = _solve(ctx, grid, Zs, rho, D.POISSON_EXACT, DFTA_POISSON_NOFOLD_LDS="1", **kv)
            assert np.array_equal(Uf.view(np.int64), Us.view(np.int64)), (L, Zs, kv)
            assert np.array_equal(vcf, vcs) and np.array_equal(ef.view(np.int64), es.view(np.int64)), (L, Zs, kv, vcf, vcs)
    grid.close()


def test_resident_groups_are_deterministic(ctx):
    """He at 16385 nodes: one shared level, short passes, the cycle stops early -- exchanges follow each other within microseconds.
    Thirty SCF steps twice: every step's U and V-cycle count identical, and identical to the one-workgroup solver's."""
    L, d, R = GRIDS["L14"]
    grid = D.Grid(ctx, L, d, R)
    runs = []
    for kv in ({}, {}, {"DFTA_POISSON_GROUP": "0"}):
        with env(**kv):
            scf = D.Scf(ctx, grid, [2], lsda=False)
            tr = []
            for _ in range(30):
                st = scf.step()
                tr.append((st.vcycles, scf.array(5, 0).copy(), scf.energies()[0][0].Etotal))
            runs.append((scf.poisson_info()[0], tr))
            scf.close()
    assert runs[0][0] == 33 and runs[1][0] == 33 and runs[2][0] == 1
    for k in range(30):
        for other in (1, 2):
            assert runs[0][1][k][0] == runs[other][1][k][0], (k, other)
            assert np.array_equal(runs[0][1][k][1].view(np.int64), runs[other][1][k][1].view(np.int64)), (k, other)
            assert runs[0][1][k][2] == runs[other][1][k][2]
    grid.close()


def test_resident_groups_are_deterministic_radon(ctx):
    """Rn at 131073 nodes (four shared levels, the hand-over to the coarse workgroup twice per V-cycle) and a batch of three
    different atoms: 10 SCF steps, twice with resident groups and once with one workgroup per atom -- U of every step identical."""
    L, d, R = GRIDS["L17"]
    grid = D.Grid(ctx, L, d, R)
    for Zs in ([86], [86, 30, 7]):
        runs = []
        for kv in ({}, {}, {"DFTA_POISSON_GROUP": "0"}):
            with env(**kv):
                scf = D.Scf(ctx, grid, Zs, lsda=False)
                tr = []
                for _ in range(10):
                    scf.step(want_stats=False)
                    tr.append([scf.array(5, a).copy() for a in range(len(Zs))])
                runs.append((scf.poisson_info()[0], tr))
                scf.close()
        assert runs[0][0] == 33 and runs[2][0] == 1
        for k in range(10):
            for a in range(len(Zs)):
                for other in (1, 2):
                    assert np.array_equal(runs[0][1][k][a].view(np.int64), runs[other][1][k][a].view(np.int64)), (Zs, k, a, other)
    grid.close()


def test_resident_lost_member_is_detected(ctx):
    L, d, R = GRIDS["L17"]
    grid = D.Grid(ctx, L, d, R)
    rr = grid.r()
    rho = (86 * np.exp(-2 * rr) / np.pi)[None, :]
    U0, vc0, _, info0 = _solve(ctx, grid, [86], rho)
    assert info0 == (33, False, 0)
    with env(DFTA_FAULT_POISSON_MEMBER="1"):
        ps = D.Poisson(ctx, grid, 1)
        U1, vc1, _ = ps.solve([86], rho)
        assert ps.group_info() == (33, True, 1)
        ps.close()
    assert np.array_equal(U0.view(np.int64), U1.view(np.int64)) and np.array_equal(vc0, vc1)
    grid.close()


@pytest.mark.parametrize("kv", [{}, {"DFTA_POISSON_NORC": "1"}, {"DFTA_POISSON_RES": "0"}, {"DFTA_POISSON_GROUP": "0"}, {"DFTA_POISSON_GROUP": "1"},
                                {"DFTA_POISSON_GROUP": "2"}, {"DFTA_POISSON_GROUP": "3"}, {"DFTA_POISSON_GROUP": "4"},
                                {"DFTA_POISSON_GROUP": "4", "DFTA_POISSON_NORC": "1"}])
def test_tolerance_mode_poisson(ctx, kv):
    """opt-in 32-node warm-ups, every flavour of the solver (the default: resident groups with the coarse workgroup's V-cycle in registers;
    POISSON_NORC: its level-by-level code): U within 2e-9 Z of the exact mode's (= the reference's) solution"""
    L, d, R = GRIDS["L17"]
    grid = D.Grid(ctx, L, d, R)
    rr = grid.r()
    worst = 0.0
    for Z in (1, 18, 86):
        rho = (Z * np.exp(-2 * rr) / np.pi)[None, :]
        Ue, vce, erre, _ = _solve(ctx, grid, [Z], rho, D.POISSON_EXACT, **kv)
        Ut, vct, errt, _ = _solve(ctx, grid, [Z], rho, D.POISSON_TOLERANCE, **kv)
        dU = float(np.max(np.abs(Ue - Ut))) / Z
        worst = max(worst, dU)
        assert dU <= 2e-9, (Z, dU)
        assert float(errt[0]) <= 10 * float(erre[0]) + 1e-13, (Z, float(errt[0]), float(erre[0]))     # the cycle ends on the same round-off floor
        # (where the cycle count is decided by the 1e-14 test -- small Z -- it is decided by round-off: not compared)
        assert 1 <= int(vct[0]) <= 100
        assert np.max(np.abs(Ut[0] - Z * (1 - (1 + rr) * np.exp(-2 * rr)))) < 3e-7 * Z     # analytic Hartree potential of 1s
    print("tolerance mode %s: max |dU| / Z = %.2e" % (kv, worst))
    grid.close()


@pytest.mark.parametrize("grid_key", ["L14", "L17"])
def test_tolerance_mode_register_cycle_in_a_batch(ctx, grid_key):
    """the register cycle of the coarse workgroup with three atoms side by side (3 x 33 workgroups) on the 16385- and the
    131073-node grid: every atom within 2e-9 Z of the exact solve, and the same bits as the atom solved alone (groups are independent)"""
    L, d, R = GRIDS[grid_key]
    grid = D.Grid(ctx, L, d, R)
    rr = grid.r()
    Zs = [1, 18, 86]
    rho = np.stack([Z * np.exp(-2 * rr) / np.pi for Z in Zs])
    Ue, _, _, info_e = _solve(ctx, grid, Zs, rho, D.POISSON_EXACT)
    Ut, vct, _, info_t = _solve(ctx, grid, Zs, rho, D.POISSON_TOLERANCE)
    assert info_t[0] == 33 and info_e[0] == 33, (info_e, info_t)          # resident groups
    for k, Z in enumerate(Zs):
        assert float(np.max(np.abs(Ue[k] - Ut[k]))) <= 2e-9 * Z, (Z, float(np.max(np.abs(Ue[k] - Ut[k]))) / Z)
        U1, vc1, _, _ = _solve(ctx, grid, [Z], rho[k:k + 1], D.POISSON_TOLERANCE)
        assert np.array_equal(U1[0].view(np.int64), Ut[k].view(np.int64)) and int(vc1[0]) == int(vct[k])
    grid.close()


@pytest.mark.parametrize("L,delta", [(14, None), (17, None), (14, 1e-6)])
def test_tolerance_mode_on_uniform_and_nearly_uniform_grids(ctx, L, delta):
    """With delta = 0 (SolvePoissonUniform) or tiny the smoother has no drift term and the smooth error modes are removed on the coarsest
    levels only: a coarse-level sweep that is not the Gauss-Seidel sweep -- e.g. a carry that does not reach every lane behind a wave
    boundary, or a stop test that fires too early on the levels with < 16 nodes -- leaves the cycle stalling at 1e-8 here while the
    logarithmic grids of the other tests still pass.  Gate as everywhere: 2e-9 Z against the exact solve, and the same end residual."""
    grid = D.Grid(ctx, L, delta, 25.0)
    rr = grid.r()
    for Z in (10, 86):
        rho = (Z * np.exp(-2 * rr) / np.pi)[None, :]
        Ue, _, erre, info = _solve(ctx, grid, [Z], rho, D.POISSON_EXACT)
        Ut, vct, errt, _ = _solve(ctx, grid, [Z], rho, D.POISSON_TOLERANCE)
        Un, _, _, _ = _solve(ctx, grid, [Z], rho, D.POISSON_TOLERANCE, DFTA_POISSON_NORC="1")
        assert info[0] == 33
        dt, dn = float(np.max(np.abs(Ue - Ut))) / Z, float(np.max(np.abs(Ue - Un))) / Z
        assert dt <= 2e-9 and dn <= 2e-9, (Z, dt, dn)
        assert float(errt[0]) <= 10 * float(erre[0]) + 1e-13, (float(errt[0]), float(erre[0]))
    grid.close()


@pytest.mark.parametrize("L,delta,R,gate", [(14, None, 25.0, 1e-8), (14, 1e-6, 25.0, 1e-8), (20, 1.25e-5, 50.0, 2e-8)])
def test_adaptive_mode_on_uniform_and_very_large_grids(ctx, L, delta, R, gate):
    """the adaptive stop rule (norm not below 0.7 x the previous cycle's twice in a row, and below 1e-3 of the first cycle's) was
    validated on the logarithmic grids of 14 / 17 levels only.  On the uniform grid (delta = 0), a nearly uniform one and at 2^20 + 1 nodes
    the cycle contracts more slowly and the solve is worse conditioned: their OWN gates -- U within 1e-8 Z resp. 2e-8 Z of the exact
    100-cycle solve (observed 4e-9 Z, 9e-9 Z: the size by which the reference's own solve moves under a 1e-12 perturbation of the density
    there, tests/golden/l20_meta.json) -- the same end residual, and a cycle count that shows the rule fired on the floor, not before it."""
    grid = D.Grid(ctx, L, delta, R)
    rr = grid.r()
    for Z in (10, 86):
        rho = (Z * np.exp(-2 * rr) / np.pi)[None, :]
        Ue, vce, erre, _ = _solve(ctx, grid, [Z], rho, D.POISSON_EXACT)
        Ua, vca, erra, _ = _solve(ctx, grid, [Z], rho, D.POISSON_ADAPTIVE)
        dU = float(np.max(np.abs(Ue - Ua))) / Z
        print("adaptive, %d levels, delta %s, Z %d: %d cycles (exact: %d), max |dU| / Z = %.2e, end residual %.2e (exact %.2e)"
              % (L, delta, Z, int(vca[0]), int(vce[0]), dU, float(erra[0]), float(erre[0])))
        assert dU <= gate, (L, delta, Z, dU)
        assert 4 <= int(vca[0]) <= 40 and int(vca[0]) <= int(vce[0]), (int(vca[0]), int(vce[0]))
        assert float(erra[0]) <= 3 * float(erre[0]) + 1e-13, (float(erra[0]), float(erre[0]))
    grid.close()


@pytest.mark.parametrize("kv", [{}, {"DFTA_POISSON_RES": "0"}, {"DFTA_POISSON_GROUP": "0"}, {"DFTA_POISSON_GROUP": "2"}])
def test_adaptive_mode_stops_on_the_round_off_floor(ctx, kv):
    """DFTA_POISSON_ADAPTIVE: the tolerance mode's kernels, and the V-cycles end where the norm of the last level-0 sweep has stopped falling
    (twice in a row not below 0.7 x the cycle before) instead of at the reference's cap of 100 -- whose own test, < 1e-14, lies below the
    floor for every Z >= 2.  The floor is reached after 6 .. 8 cycles (30x per cycle, then flat); what the remaining 92
    cycles of the reference do is wander by < 1e-9 Z.  Gates: U within 2e-9 Z of the exact 100-cycle solve, the same end residual, at
    most 12 cycles; Z = 1, where the reference's test IS met (80 cycles), included."""
    for key in ("L14", "L17"):
        L, d, R = GRIDS[key]
        grid = D.Grid(ctx, L, d, R)
        rr = grid.r()
        for Z in (1, 18, 86):
            rho = (Z * np.exp(-2 * rr) / np.pi)[None, :]
            Ue, vce, erre, _ = _solve(ctx, grid, [Z], rho, D.POISSON_EXACT, **kv)
            Ua, vca, erra, _ = _solve(ctx, grid, [Z], rho, D.POISSON_ADAPTIVE, **kv)
            dU = float(np.max(np.abs(Ue - Ua))) / Z
            assert dU <= 2e-9, (key, Z, dU)
            assert 3 <= int(vca[0]) <= 12 and int(vca[0]) <= int(vce[0]), (key, Z, int(vca[0]), int(vce[0]))
            assert float(erra[0]) <= 2 * float(erre[0]) + 1e-13, (key, Z, float(erra[0]), float(erre[0]))
        grid.close()


def test_tolerance_mode_poisson_at_a_million_nodes(ctx):
    """1 048 577 nodes (staged group of 32 workgroups, workgroup 0's levels from 8193 nodes down in registers; POISSON_NORC: level by level).
    At this size the solve is conditioned like 1e-8 -- the compiled reference's own U moves by max |dU| = 5.4e-7 when the density is
    perturbed by 1e-12 relative (tests/golden/l20_meta.json, make_golden_table.py l20cond) -- so the gate against the exact solve (which
    returns the reference's bits here, test_gpu_configs.py) is 4x that reference-measured figure, not the 2e-9 Z of the smaller grids."""
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "l20_meta.json")) as f:
        ref_abs = json.load(f)["poisson_conditioning"]["max_abs_dU"]
    L, d, R = GRIDS["L20"]
    grid = D.Grid(ctx, L, d, R)
    rr = grid.r()
    Z = 86
    rho = (Z * np.exp(-2 * rr) / np.pi)[None, :]
    Ue, _, _, info = _solve(ctx, grid, [Z], rho, D.POISSON_EXACT)
    Ut, vct, _, _ = _solve(ctx, grid, [Z], rho, D.POISSON_TOLERANCE)
    Un, _, _, _ = _solve(ctx, grid, [Z], rho, D.POISSON_TOLERANCE, DFTA_POISSON_NORC="1")
    assert info[0] == 32
    dt, dn = float(np.max(np.abs(Ue - Ut))), float(np.max(np.abs(Ue - Un)))
    print("tolerance mode at 2^20+1 nodes: max |dU| %.2e (register cycle), %.2e (level by level); the reference's conditioning %.2e" % (dt, dn, ref_abs))
    assert dt <= 4 * ref_abs and dn <= 4 * ref_abs
    assert np.max(np.abs(Ut[0] - Z * (1 - (1 + rr) * np.exp(-2 * rr)))) < 3e-6 * Z     # analytic Hartree potential of 1s
    assert 1 <= int(vct[0]) <= 100
    grid.close()


@pytest.mark.parametrize("lsda", [False, True])
def test_tolerance_mode_radon_steps_vs_reference(ctx, lsda):
    """BASELINE configs[1] / [2] in tolerance mode against the compiled reference's golden steps: energies 1e-9 relative,
    eigenvalues 1e-8 Ha + 2e-9 |E| (the gate of every step that has been through a Poisson solve, test_gpu_configs.py)"""
    L, d, R = GRIDS["L17"]
    grid = D.Grid(ctx, L, d, R)
    gold = json.load(open(os.path.join(HERE, "golden", "rn_end_to_end.json")))["Rn_LSDA_L17" if lsda else "Rn_LDA_L17"]
    scf = D.Scf(ctx, grid, [86], lsda=lsda, levels_mode=D.LEVELS_CHAINED, poisson_mode=D.POISSON_TOLERANCE)
    worst_e, worst_l = 0.0, 0.0
    for key in ("first", "second"):
        scf.step()
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
    # and to the end of the recorded trajectory: Etotal of every step
    traj = np.array(gold["etotal_all"])
    scf2 = D.Scf(ctx, grid, [86], lsda=lsda, poisson_mode=D.POISSON_TOLERANCE)
    got = []
    for _ in range(len(traj)):
        scf2.step(want_stats=False)
        got.append(scf2.energies()[0][0].Etotal)
    rel = np.abs(np.array(got) - traj) / np.abs(traj)
    print("tolerance mode Rn %s: steps 0/1 energies %.2e rel, eigenvalues %.2e |E|; trajectory (%d steps) %.2e"
          % ("LSDA" if lsda else "LDA", worst_e, worst_l, len(traj), rel.max()))
    assert rel.max() <= 2e-9
    scf.close()
    scf2.close()
    grid.close()
