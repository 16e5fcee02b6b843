#!/usr/bin/env python3
"""Rounds of the level search per SCF step for atoms other than Rn (are the noise guards / kappa tuned on Rn's
fifteen levels Rn-specific?): one atom at a time, LDA (and LSDA for two of them), 131073 nodes, SCF steps 5..24.
    python profiles/rounds_other_atoms.py > profiles/r03_rounds_other_atoms.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import dftatom_amd as D  # noqa: E402

ctx = D.Context(0)
grid = D.Grid(ctx, 17, 1e-4, 50.0)
out = {"what": "level-search rounds per SCF step, steps 5..24, one atom, 131073 nodes", "atoms": {}}
for Z, lsda in ((10, False), (26, False), (47, False), (64, False), (79, False), (86, False), (26, True), (64, True), (86, True)):
    scf = D.Scf(ctx, grid, [Z], lsda=lsda)
    r, ms, iu = [], [], []
    for it in range(25):
        st = scf.step()
        if it >= 5:
            r.append(st.rounds); ms.append(st.ms_levels); iu.append(st.sweeps_issued / max(st.sweeps_reference_executed, 1))
    out["atoms"]["Z=%d %s" % (Z, "LSDA" if lsda else "LDA")] = {"jobs": scf.njobs, "rounds_mean": sum(r) / len(r), "rounds_min": min(r), "rounds_max": max(r),
                                                              "ms_levels_mean": sum(ms) / len(ms), "issued_per_useful": sum(iu) / len(iu)}
    scf.close()
print(json.dumps(out, indent=1))
