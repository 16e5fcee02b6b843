"""CPU suite: host logic of the periodic-table sweep -- static atom partition and the one collective
(all_gather of fixed-size records), exercised with world_size 2 over gloo."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dftatom_amd import sweep


def test_partition_is_balanced_and_complete():
    Zs = list(range(1, 87))
    assert sum(sweep.subshell_count(z) for z in Zs) == 814
    for world in (1, 2, 4, 8):
        # by work (subshells x expected SCF steps): LPT keeps the ranks within one atom's cost of each other
        shards = sweep.partition_atoms(Zs, world, cost=sweep.atom_cost)
        assert sorted(z for s in shards for z in s) == Zs
        loads = [sum(sweep.atom_cost(z) for z in s) for s in shards]
        assert max(loads) - min(loads) <= max(sweep.atom_cost(z) for z in Zs)
        assert max(loads) <= 1.03 * sum(loads) / world + 1
        by_jobs = sweep.partition_atoms(Zs, world, cost=sweep.subshell_count)     # the round-1 weight is still available
        assert sorted(z for s in by_jobs for z in s) == Zs
        # default: balance of the PREDICTED SHARD TIMES (critical path + work): complete, and never predicted slower than by-work LPT
        model = sweep.partition_atoms(Zs, world)
        assert sorted(z for s in model for z in s) == Zs
        t_model = max(sweep.shard_time_ms(s) for s in model)
        t_work = max(sweep.shard_time_ms(s) for s in shards)
        assert t_model <= 1.02 * t_work, (world, t_model, t_work)
    assert sweep.partition_atoms(Zs, 8) == sweep.partition_atoms(Zs, 8)   # deterministic on every rank
    # a sweep without cap-hitting atoms on every rank: the model keeps the long runners together so that the other ranks finish early
    some = [2, 10, 18, 36, 54, 86, 3, 4, 6, 7, 12, 13]
    parts = sweep.partition_atoms(some, 4)
    assert sorted(z for s in parts for z in s) == sorted(some)
    # the Z = 87..118 extension of the table (round 3)
    assert sweep.expected_steps(118) == 60 and sweep.expected_steps(87) == 100 and sweep.expected_steps(119) == 100


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shards = sweep.partition_atoms(list(range(1, 21)), world)
    cap = max(len(s) for s in shards)
    rec = np.zeros((len(shards[rank]), sweep.RECORD_DOUBLES))
    for k, z in enumerate(shards[rank]):                # synthetic records: Z, Etotal = -Z^2.4, ..., eigenvalues
        rec[k, 0], rec[k, 1], rec[k, 6], rec[k, 7], rec[k, 8], rec[k, 9] = z, -float(z) ** 2.4, 1, 30 + rank, 2, 1
        rec[k, 10:12] = [-z * z / 2.0, -z * z / 8.0]
    table = sweep.gather_records(torch.from_numpy(sweep.pack_records(rec, cap)), dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, sorted(table), [table[z][1] for z in sorted(table)]))


def test_gather_world_size_2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, zs, et in res:                              # every rank ends with the full table
        assert zs == list(range(1, 21))
        assert np.allclose(et, [-float(z) ** 2.4 for z in zs])
    f = sweep.record_fields(np.concatenate([[3, -7.3, 7.2, 4.0, -17.0, -1.5, 1, 31, 2, 1, -1.9, -0.08], np.zeros(52)]))
    assert f["Z"] == 3 and f["finished"] and f["nlevels"] == 2 and f["eigenvalues"].tolist() == [-1.9, -0.08]


def _worker8(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    shards = sweep.partition_atoms(list(range(1, 87)), world)
    cap = max(len(s) for s in shards)
    rec = np.zeros((len(shards[rank]), sweep.RECORD_DOUBLES))
    for k, z in enumerate(shards[rank]):                # fake records: Z, Etotal, finished, steps (= the rank that made it), levels
        rec[k, 0], rec[k, 1], rec[k, 6], rec[k, 7], rec[k, 8], rec[k, 9] = z, -float(z) ** 2.4, 1, rank, 1, 1
        rec[k, 10] = -z * z / 2.0
    table = sweep.gather_records(torch.from_numpy(sweep.pack_records(rec, cap)), dist)
    seen = dist.get_world_size()
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, seen, [len(s) for s in shards], sorted(table), {int(z): int(table[z][7]) for z in table}))


def test_periodic_table_partition_and_gather_world_size_8_gloo():
    """BASELINE config 4 as the driver will launch it on an 8-GPU node, minus the GPUs: eight ranks over gloo, the
    partition of Z = 1..86 complete and disjoint and the same on every rank, every rank packs the records of its shard, ONE all_gather
    returns all 86 to everybody, each made by the rank that owns the atom"""
    shards = sweep.partition_atoms(list(range(1, 87)), 8)
    assert sorted(z for s in shards for z in s) == list(range(1, 87)) and sum(len(s) for s in shards) == 86      # complete, disjoint
    assert all(len(s) >= 1 for s in shards)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    owner = {z: r for r, sh in enumerate(shards) for z in sh}
    assert sorted(r[0] for r in res) == list(range(8))
    for rank, seen, per_rank, zs, made_by in res:
        assert seen == 8 and per_rank == [len(sh) for sh in shards] and sum(per_rank) == 86      # the same partition on every rank
        assert zs == list(range(1, 87))                                                          # every rank ends with the full table
        assert made_by == owner


def test_shard_time_model_reproduces_the_recorded_shards():
    """A CONSISTENCY check, not a validation: the constants of sweep.shard_time_ms are a least-squares fit to the 15 recorded
    shards of the emulated 1-, 2-, 4-, 8-rank sweeps (profiles/fit_shard_model.py, profiles/<newest round>_periodic_table_predicted_scaling_<mode>.json),
    and this test checks that the constants in the source ARE that fit (predictions equal to 5 %) and that the fit describes the shards it
    was made from (15 %).  It is in-sample: the three features are nearly collinear, and fitted on the 1-, 2- and 4-rank shards alone the
    model misses the 8-rank shards by up to 21 % (exact kernels) / 70 % (tolerance modes) -- it interpolates the recorded partitions, it
    does not extrapolate to other rank counts or kernels.  The partition it drives is only a load-balancing heuristic; results never
    depend on it (tests/test_gpu_compat.py::test_periodic_table_two_ranks_equal_one_rank), and the measured number of config 4 is
    bench.py's `extra.periodic_table` (this run's ranks), not this model."""
    import importlib.util
    import json
    import os
    import numpy as np
    from dftatom_amd import sweep
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fit_shard_model", os.path.join(root, "profiles", "fit_shard_model.py"))
    fitm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fitm)
    for mode in ("exact", "tolerance"):
        with open(fitm.path_of(mode)) as f:
            rec = json.load(f)
        n = 0
        for run in rec["runs"]:
            for sh in run["shards"]:
                if not sh["atoms"]:
                    continue
                pred = sweep.shard_time_ms(sh["atoms"], mode) / 1e3
                assert abs(pred - sh["seconds"]) <= 0.15 * sh["seconds"], (mode, run["emulated_ranks"], sh["rank"], pred, sh["seconds"])
                n += 1
        assert n == 15
        coef, rel = fitm.fit(mode)
        # the constants in the source are this fit: the three features are nearly collinear over 15 shards, so the constants themselves
        # wobble from one emulation to the next -- what must agree are the PREDICTIONS (5 % on every recorded shard)
        for run in rec["runs"]:
            for sh in run["shards"]:
                if sh["atoms"]:
                    a = float(np.dot(coef, sweep.shard_features(sh["atoms"])))
                    b = sweep.shard_time_ms(sh["atoms"], mode)
                    assert abs(a - b) <= 0.05 * a, (mode, run["emulated_ranks"], sh["rank"], a, b)
        assert float(np.max(np.abs(rel))) <= 0.15
    # the prediction itself: an eighth of the table per GPU
    with open(fitm.path_of("tolerance")) as f:
        tol = json.load(f)
    assert tol["predicted_seconds"]["8"] <= 7.0            # (reads the recorded file: the newest emulation, a PREDICTION from one GPU)
