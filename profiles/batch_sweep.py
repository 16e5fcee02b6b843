#!/usr/bin/env python3
"""Batch-size sweep of the throughput regime: B identical Rn atoms (LDA, 131073 nodes) advanced together,
B = 64 .. 2048, >= 10 timed SCF steps after >= 5 warm-up steps (so that the history predictions of the level search are active).
Per batch size: executed and reference-equivalent sweeps/s, V-cycles/s, issued trials per useful one, rounds per step, ms per
atom-step and the per-kernel roofline fractions -- where the throughput saturates.

    python profiles/batch_sweep.py [--sizes 64,128,256,512,1024,2048] [--steps 10] [--warmup 5] > profiles/r03_batch_sweep.json
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="64,128,256,512,1024,2048")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scan-sweeps", action="store_true", help="the sweeps' tolerance mode (scan.hip)")
    ap.add_argument("--tolerance", action="store_true", help="the multigrid's tolerance mode")
    args = ap.parse_args()
    import torch
    import bench
    import dftatom_amd as D
    ctx = D.Context(0, torch.cuda.current_stream().cuda_stream)
    grid = D.Grid(ctx, 17, 1e-4, 50.0)
    bench.HBM_MEASURED["copy"], bench.HBM_MEASURED["triad"] = ctx.measure_hbm(1 << 27, 5)
    out = {"workload": "B Rn atoms, LDA, 131073 nodes, %d steps after %d warm-up steps; sweeps %s, multigrid %s"
                       % (args.steps, args.warmup, "tolerance (scan)" if args.scan_sweeps else "exact", "tolerance" if args.tolerance else "exact"),
           "hbm_copy_GBps": bench.HBM_MEASURED["copy"], "sizes": {}}

    def barrier():
        torch.cuda.synchronize()

    for B in [int(x) for x in args.sizes.split(",")]:
        t0 = time.time()
        scf, tot = bench.run_workload(D, ctx, grid, 17, B, False, args.steps, args.warmup, 0, barrier, torch,
                                      poisson_mode=D.POISSON_TOLERANCE if args.tolerance else None, sweep_mode=D.SWEEPS_TOLERANCE if args.scan_sweeps else None)
        scf.close()
        s = bench.summarize(tot, 17, grid.N, B, False, 1, 1e-4, 50.0)
        out["sizes"][str(B)] = {"ms_per_step": s["ms_per_step"], "ms_per_atom_step": s["ms_per_atom_step"],
                                "sweeps_executed_per_s": s["sweeps_executed_per_s"],
                                "sweeps_reference_equivalent_per_s": s["sweeps_reference_equivalent_per_s"],
                                "vcycles_per_s": s["vcycles_per_s"], "issued_per_useful": s["issued_per_useful"],
                                "rounds_per_step": s["rounds_per_step"], "phase_ms_per_step": s["phase_ms_per_step"],
                                "tree_depth": tot["tree_depth"], "poisson_workgroups_per_atom": tot["poisson_G"],
                                "sweep_kernel": s["kernels"]["sweep"]["kernel"].split(" ")[0], "levels_ms": s["phase_ms_per_step"]["levels"],
                                "poisson_ms": s["phase_ms_per_step"]["poisson"],
                                "sweep_frac_issued": s["kernels"]["sweep"]["frac_issued"],
                                "sweep_frac_executed_path": s["kernels"]["sweep"]["frac"],
                                "poisson_frac_algorithmic": s["kernels"]["poisson"]["frac"],
                                "wall_s": time.time() - t0}
        sys.stderr.write("B=%d: %.1f ms/step, %.0f sweeps/s, issued/useful %.2f, rounds %.1f\n"
                         % (B, s["ms_per_step"], s["sweeps_executed_per_s"], s["issued_per_useful"], s["rounds_per_step"]))
    print(json.dumps(out, indent=1))
    grid.close()
    ctx.close()


if __name__ == "__main__":
    main()
