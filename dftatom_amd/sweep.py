"""Host-side sharding of independent atoms over ranks (periodic-table sweep, BASELINE.json config 4).

Atoms are independent SCF problems: there is no exchange during the solve.  Ranks get a
static partition that balances the predicted wall time of the shards (critical path = steps of the slowest atom x the
latency floor of a step, plus work = subshells x expected SCF steps) and the only collective is one all_gather of
fixed-size result records (RECORD_DOUBLES doubles per atom) at the end.  The same code runs over RCCL (backend "nccl", GPU tensors) and, in the CPU tests, over gloo.
"""
import numpy as np

RECORD_DOUBLES = 64


def subshell_count(Z):
    """Number of occupied subshells (AufbauPrinciple.h:36-75): the per-step Numerov job count of an atom."""
    from . import get_subshells
    return len(get_subshells(Z))


# SCF steps until the stop test |dE/E| < 1e-11 held in two consecutive steps (or the cap of 100, DFTAtom.cpp:396), Z = 1..118,
# LDA, 131073 nodes, mixing 0.5: what the compiled reference took (tests/golden/periodic_table_L17.json).  The stop step is
# round-off noise in its last digits, but which atoms are slow (open shells: up to the cap) is not.
EXPECTED_STEPS = (100, 100, 81, 68, 100, 64, 51, 100, 65, 38, 100, 30, 36, 57, 100, 44, 100, 68, 44, 68,
                  31, 64, 66, 51, 100, 76, 99, 41, 100, 100, 59, 42, 55, 78, 100, 89, 100, 34, 43, 100,
                  95, 31, 100, 56, 97, 100, 49, 83, 100, 44, 32, 88, 97, 81, 100, 35, 62, 65, 42, 93,
                  51, 77, 34, 27, 42, 76, 63, 100, 100, 100, 100, 35, 100, 47, 69, 100, 67, 31, 33, 100,
                  60, 100, 32, 58, 27, 35, 100, 45, 85, 100, 100, 31, 62, 95, 35, 61, 46, 100, 45, 39,
                  100, 100, 92, 95, 51, 30, 93, 100, 45, 100, 100, 95, 100, 100, 77, 36, 75, 60)


def expected_steps(Z):
    return EXPECTED_STEPS[Z - 1] if 1 <= Z <= len(EXPECTED_STEPS) else 100


def atom_cost(Z):
    """Work of one atom of a sweep: Numerov jobs per SCF step x expected SCF steps (a finished atom is frozen and costs nothing
    more, dfta_scf_step), SURVEY.md section 8e."""
    return subshell_count(Z) * expected_steps(Z)


# Time model of one shard (measured on MI355X, profiles/r06_periodic_table_predicted_scaling*.json; round 4's constants were 62.5 / 69.4 / 0.221 and 10.8 / 24.0 / 0.144): a step of a batch costs a
# latency floor -- the level search and the multigrid's dependent sweeps take what they take for one atom or ten -- plus a per-job
# share once the batch fills the machine:  t_step = FLOOR(live atoms) + JOB_MS x (subshells of the atoms still running).  The floor
# has two values: with up to RESIDENT_MAX_ATOMS live atoms the multigrid runs in resident groups (and, in tolerance mode, the coarse
# workgroup's V-cycle in registers and the level search in groups of workgroups per level), above that in staged groups.  A shard runs
# until its slowest atom stops, so  T(shard) = FLOOR_SMALL_MS x (steps with <= 7 live atoms) + FLOOR_BIG_MS x (steps with more) +
# JOB_MS x sum(subshells x steps): the first two terms are the critical path (what a by-work partition ignores), the third the work.
# One triple per mode of the sweeps, least squares over the shards of the emulated 1-, 2-, 4- and 8-rank sweeps (profiles/fit_shard_model.py
# re-fits them from the recorded files; tests/test_sweep_dist.py checks that the model reproduces every recorded shard time within 15 %).
RESIDENT_MAX_ATOMS = 7
SHARD_MODEL = {"exact": (44.9, 51.3, 0.233),        # round-6 kernels incl. the 17-workgroup resident multigrid of 8 .. 15 atoms and the device-side search of up to 256 levels (profiles/r06_periodic_table_predicted_scaling_exact.json): residuals of the 15 recorded shards max 5.0 %, rms 2.9 %
               "tolerance": (5.6, 11.4, 0.153)}     # scan sweeps + the multigrid's tolerance mode (..._tolerance.json): max 11.1 %, rms 5.5 %
STEP_FLOOR_MS, JOB_MS = SHARD_MODEL["exact"][1], SHARD_MODEL["exact"][2]


def shard_features(Zs):
    """(steps with <= RESIDENT_MAX_ATOMS live atoms, steps with more, sum of subshells x steps) of a shard, from the expected step counts"""
    steps = [expected_steps(z) for z in Zs]
    smax = max(steps)
    small = sum(1 for k in range(1, smax + 1) if sum(1 for t in steps if t >= k) <= RESIDENT_MAX_ATOMS)
    return small, smax - small, sum(atom_cost(z) for z in Zs)


def shard_time_ms(Zs, model="exact"):
    """predicted wall time of one rank that advances the atoms Zs together until each has stopped (see the model above)"""
    if not Zs:
        return 0.0
    f_small, f_big, job = SHARD_MODEL[model]
    small, big, work = shard_features(Zs)
    return f_small * small + f_big * big + job * work


def partition_atoms(Zs, world_size, cost=None, model="exact"):
    """Static assignment of atoms to ranks, deterministic on every rank: returns a list (per rank) of lists of Z.

    Default (cost=None): greedy on the PREDICTED SHARD TIME (critical path + work, shard_time_ms): atoms in order of decreasing step
    count, each to the rank whose predicted time grows least / stays smallest.  Long runners (the atoms that hit the reference's
    100-step cap) therefore land together on as few ranks as the work balance allows -- their late steps keep some batch width and
    the other ranks finish early instead of every rank idling through 100 latency-bound steps -- while the work term keeps the
    ranks balanced.  cost=callable: plain longest-processing-time on that additive cost (round 2's subshells x steps)."""
    if cost is not None:
        order = sorted(Zs, key=lambda z: (-cost(z), z))
        loads = [0] * world_size
        shards = [[] for _ in range(world_size)]
        for z in order:
            r = min(range(world_size), key=lambda k: (loads[k], k))
            shards[r].append(z)
            loads[r] += cost(z)
        return [sorted(s) for s in shards]
    order = sorted(Zs, key=lambda z: (-expected_steps(z), -atom_cost(z), z))
    shards = [[] for _ in range(world_size)]
    for z in order:
        best = min(range(world_size), key=lambda k: (max(shard_time_ms(shards[j] + ([z] if j == k else []), model) for j in range(world_size)),
                                                     shard_time_ms(shards[k] + [z], model), k))
        shards[best].append(z)
    return [sorted(s) for s in shards]


def pack_records(records, capacity):
    """records: array (n, RECORD_DOUBLES) -> fixed (capacity, RECORD_DOUBLES) block, unused rows have Z = 0."""
    out = np.zeros((capacity, RECORD_DOUBLES))
    rec = np.asarray(records, dtype=np.float64).reshape(-1, RECORD_DOUBLES)
    out[: len(rec)] = rec
    return out


def gather_records(local_block, dist=None):
    """all_gather fixed-size record blocks (torch tensors, CPU for gloo / CUDA for RCCL); returns dict Z -> record."""
    import torch
    t = local_block if isinstance(local_block, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(local_block))
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        blocks = [t]
    else:
        blocks = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(blocks, t)
    table = {}
    for b in blocks:
        for row in b.detach().cpu().numpy().reshape(-1, RECORD_DOUBLES):
            if row[0] > 0:
                table[int(row[0])] = row.copy()
    return table


def record_fields(row):
    """Decode one record written by k_energies (scf.hip)."""
    nlev = int(row[8])
    return {"Z": int(row[0]), "Etotal": row[1], "Ekinetic": row[2], "Ecoul": row[3], "Enuclear": row[4], "Exc": row[5],
            "finished": bool(row[6]), "steps": int(row[7]), "nlevels": nlev, "converged": bool(row[9]),
            "eigenvalues": row[10:10 + nlev].copy()}
