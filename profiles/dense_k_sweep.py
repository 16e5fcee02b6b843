#!/usr/bin/env python3
"""Isolated sweep-kernel benchmark of SURVEY.md section 8(d): the synthetic DENSE trial set.

For every occupied (n, l) level of Rn (15 subshells) K energies uniformly spaced in [-Z^2 - 1, 50], K in {64, 512}, on the SCF
potential of step 0 (the reference's flat start density, DFTAtom.cpp:371-392): K x 15 CountNodes sweeps (limit = the level's node
count, DFTAtom.cpp:497) and K x 15 SolutionInZero sweeps per launch, through the C ABI (dfta_numerov_sweeps: host boundary values,
bit-identical to the reference's sweeps).  Reported per launch: kernel time (HIP events around the sweep kernel inside the
library), sweeps/s, traversed points/s and the SURVEY-8d roofline figure 8 B x traversed points / time against 8 TB/s and against
the measured copy bandwidth.  No speculation here: every trial is useful work, so this is the kernel's own ceiling for the level
search (a search issues trees of such trials).

    python profiles/dense_k_sweep.py            # prints one JSON object (also part of bench.py's `extra`)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(D, ctx, grid, peak_gbs=8000.0, measured_gbs=None, Z=86, reps=3, scan=False):
    scf = D.Scf(ctx, grid, [Z], lsda=False)
    V = scf.array(3, 0).copy()                    # potential of spin 0 at step 0
    scf.close()
    levels = D.get_subshells(Z)
    out = {"workload": "Rn Z=86 step-0 potential, %d nodes, 15 subshells x K uniformly spaced energies in [-Z^2-1, 50]" % grid.N, "sets": {}}
    for K in (64, 512):
        E = np.tile(np.linspace(-float(Z) * Z - 1.0, 50.0, K), len(levels))
        l = np.repeat([lv[1] for lv in levels], K)
        lim = np.repeat([lv[0] - lv[1] for lv in levels], K)        # NumNodes = m_N - m_L (DFTAtom.cpp:497)
        for kind, name in ((D.SWEEP_COUNT, "count_nodes"), (D.SWEEP_ZERO, "solution_in_zero")):
            best, pts = None, 0
            for _ in range(reps + 1):
                if scan:          # tolerance mode: one workgroup per trial (dfta_numerov_sweeps_scan); trips = the turning-point exits
                    r = D.numerov_sweeps_scan(ctx, grid, kind, V, l, E, lim if kind == D.SWEEP_COUNT else None)
                else:
                    r = D.numerov_sweeps(ctx, grid, kind, V, l, E, lim if kind == D.SWEEP_COUNT else None)
                ms = ctx.last_kernel_ms()
                pts = int(r["trip"].astype(np.int64).sum())
                best = ms if best is None else min(best, ms)
            nsw = len(E)
            gbs = 8.0 * pts / (best * 1e-3) / 1e9
            out["sets"]["K%d_%s" % (K, name)] = {
                "sweeps": nsw, "kernel_ms": best, "sweeps_per_s": nsw / (best * 1e-3), "points_traversed": pts,
                "points_per_s": pts / (best * 1e-3), "achieved_GBps_8B_per_point": gbs, "frac_of_8TBps": gbs / peak_gbs,
                "frac_of_measured_copy": gbs / measured_gbs if measured_gbs else None,
                "blocks_of_64_trials": (nsw + 63) // 64}
    return out


if __name__ == "__main__":
    import torch          # noqa: F401  (one HIP runtime per process, as in bench.py)
    import dftatom_amd as D
    ctx = D.Context(0)
    grid = D.Grid(ctx, 17, 1e-4, 50.0)
    copy, triad = ctx.measure_hbm(1 << 27, 5)
    res = run(D, ctx, grid, 8000.0, copy)
    res["scan_sweeps"] = run(D, ctx, grid, 8000.0, copy, scan=True)["sets"]      # the same trial sets through the tolerance-mode kernel
    res["hbm_copy_GBps"], res["hbm_triad_GBps"] = copy, triad
    print(json.dumps(res, indent=1))
    grid.close()
    ctx.close()
