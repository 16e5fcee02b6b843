"""Periodic-table sweep (BASELINE.json config 4): LDA ground states of Z = 1 .. 86 on the 131073-node grid.

Atoms are independent: every rank takes its shard of `dftatom_amd.sweep.partition_atoms` (longest-processing-time on
subshells x expected SCF steps), advances all its atoms together (one `Scf` batch: every kernel works on the whole shard)
until each has met the reference's stop test or `--max-steps`; an atom that has finished is frozen in its own stop state and
costs nothing more (dfta_scf_step).  The fixed-size result records are gathered once (RCCL all_gather under torch.distributed).

    python examples/periodic_table.py                       # one GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/periodic_table.py
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# total energies of the NIST LDA reference tables cited by the reference's README (Hartree)
NIST_LDA = {2: -2.834836, 10: -128.233481, 18: -525.946195, 36: -2750.147940, 54: -7228.856107, 86: -21861.346869}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--zmin", type=int, default=1)
    ap.add_argument("--zmax", type=int, default=86)
    ap.add_argument("--levels", type=int, default=17)
    ap.add_argument("--max-steps", type=int, default=100, help="the reference's cap (DFTAtom.cpp:396)")
    ap.add_argument("--out", default="")
    ap.add_argument("--partition", choices=("model", "work"), default="model",
                    help="model: balance the predicted shard times (critical path + work, dftatom_amd.sweep); work: LPT on subshells x steps")
    ap.add_argument("--sweeps", choices=("exact", "tolerance"), default="exact", help="tolerance: the scan sweeps (DFTA_SWEEPS_TOLERANCE)")
    ap.add_argument("--poisson", choices=("exact", "tolerance", "adaptive"), default="exact",
                    help="tolerance: the multigrid's tolerance mode; adaptive: that, and the V-cycles stop at the round-off floor")
    ap.add_argument("--emulate-ranks", type=int, default=0,
                    help="one GPU, no launcher: run each of the N shards of an N-rank sweep alone, one after the other, and report the "
                         "per-shard wall times; their maximum PREDICTS the N-GPU wall time (shards never interact; the only collective "
                         "is a gather of 64 doubles per atom)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import dftatom_amd as D
    from dftatom_amd import sweep

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # DFTA_BENCH_SHARED_GPU=1 (tests on a one-GPU box): every rank on device 0, records gathered over gloo from host memory
    shared = world > 1 and os.environ.get("DFTA_BENCH_SHARED_GPU") == "1"
    if shared:
        local = 0
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    elif world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    ctx = D.Context(local, torch.cuda.current_stream().cuda_stream)
    delta, rmax = {14: (5e-4, 25.0), 17: (1e-4, 50.0)}.get(args.levels, (1e-4, 50.0))
    grid = D.Grid(ctx, args.levels, delta, rmax)

    Zs = list(range(args.zmin, args.zmax + 1))
    cost = sweep.atom_cost if args.partition == "work" else None
    modes = dict(sweep_mode=D.SWEEPS_TOLERANCE if args.sweeps == "tolerance" else D.SWEEPS_EXACT,
                 poisson_mode={"tolerance": D.POISSON_TOLERANCE, "adaptive": D.POISSON_ADAPTIVE}.get(args.poisson, D.POISSON_EXACT))
    model = "tolerance" if args.sweeps == "tolerance" else "exact"
    if cost is None and model != "exact":
        cost_model = model
    else:
        cost_model = "exact"
    if args.emulate_ranks > 0:
        # the N shards of an N-rank sweep, one at a time on this GPU
        assert world == 1, "--emulate-ranks runs without a launcher"
        N = args.emulate_ranks
        shards = sweep.partition_atoms(Zs, N, cost=cost, model=cost_model)
        rows = []
        for r, zs in enumerate(shards):
            t0 = time.time()
            scf = D.Scf(ctx, grid, zs, lsda=False, **modes)
            steps = 0
            while steps < args.max_steps:
                scf.step(want_stats=False)
                steps += 1
                _, fin = scf.energies()
                if fin.all():
                    break
            ctx.synchronize()
            dt = time.time() - t0
            en, fin = scf.energies()
            scf.close()
            rows.append({"rank": r, "atoms": zs, "steps": steps, "seconds": dt, "finished": int(fin.sum()),
                         "predicted_seconds_model": sweep.shard_time_ms(zs, cost_model) / 1e3,
                         "etotal": {int(z): en[k].Etotal for k, z in enumerate(zs)}})
            print("shard %d/%d: %2d atoms, %3d steps, %.2f s (model %.2f s)" % (r, N, len(zs), steps, dt, rows[-1]["predicted_seconds_model"]), flush=True)
        pred = max(x["seconds"] for x in rows)
        res = {"emulated_ranks": N, "partition": args.partition, "sweeps": args.sweeps, "poisson": args.poisson, "levels": args.levels, "zmin": args.zmin, "zmax": args.zmax,
               "predicted_n_gpu_seconds": pred, "sum_of_shard_seconds": sum(x["seconds"] for x in rows), "shards": rows,
               "note": "PREDICTION from one GPU: every shard run alone; max over shards = wall time of an N-rank sweep up to the final "
                       "all_gather of 64 doubles per atom (microseconds)"}
        print("emulated %d ranks: predicted wall time %.2f s (slowest shard), sum of shards %.2f s" % (N, pred, res["sum_of_shard_seconds"]))
        if args.out:
            with open(args.out, "w") as f:
                json.dump(res, f, indent=1)
        grid.close()
        ctx.close()
        return
    mine = sweep.partition_atoms(Zs, world, cost=cost, model=cost_model)[rank]
    cap = max(len(s) for s in sweep.partition_atoms(Zs, world, cost=cost, model=cost_model))
    t0 = time.time()
    scf = D.Scf(ctx, grid, mine, lsda=False, **modes)
    steps = 0
    while steps < args.max_steps:
        scf.step(want_stats=False)
        steps += 1
        _, fin = scf.energies()
        if fin.all():
            break
    block = torch.zeros((cap, D.RECORD_DOUBLES), dtype=torch.float64, device="cuda")
    scf.records_into(block.data_ptr())          # rows beyond len(mine) stay zero (Z = 0: no atom)
    ctx.synchronize()
    table = sweep.gather_records(block.cpu() if shared else block, dist if world > 1 else None)
    elapsed = time.time() - t0
    if rank == 0:
        rows = [sweep.record_fields(table[z]) for z in sorted(table)]
        for r in rows:
            ref = NIST_LDA.get(r["Z"])
            print("Z %3d  Etotal %16.6f  steps %3d  finished %d%s" % (r["Z"], r["Etotal"], r["steps"], r["finished"],
                  "   NIST LDA %.6f (diff %.1e)" % (ref, r["Etotal"] - ref) if ref else ""))
        print("%d atoms, %d GPU(s), %d SCF steps of the longest-running atom of rank 0, %d atom-steps in all, %d finished, %.1f s"
              % (len(rows), world, steps, sum(r["steps"] for r in rows), sum(r["finished"] for r in rows), elapsed))
        if args.out:
            with open(args.out, "w") as f:
                json.dump({"n_gpus": world, "levels": args.levels, "steps": steps, "seconds": elapsed,
                           "atoms": [{"Z": r["Z"], "Etotal": r["Etotal"], "steps": r["steps"], "finished": r["finished"]} for r in rows]},
                          f, indent=1)
    scf.close()
    grid.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
