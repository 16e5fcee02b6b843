import sys, os, time, subprocess
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import torch
import dftatom_amd as D
from golden.make_golden import GRIDS, screened_potential
ctx = D.Context(0)
for tag in ("L14", "L17"):
    L, d, R = GRIDS[tag]
    grid = D.Grid(ctx, L, d, R)
    V = screened_potential(grid.r(), 86.0)
    P = D.Potential(ctx, grid, V)
    l = np.array([0], np.int32); E = np.array([-100.0]); lim = np.array([3], np.int32)
    for name, fn in (("numerov_sweeps COUNT (upload + tables per call)", lambda: D.numerov_sweeps(ctx, grid, D.SWEEP_COUNT, V, l, E, lim)),
                     ("potential COUNT exact", lambda: (P.update(V), P.sweeps(D.SWEEP_COUNT, l, E, lim))),
                     ("potential ZERO exact", lambda: (P.update(V), P.sweeps(D.SWEEP_ZERO, l, E))),
                     ("potential COUNT scan", lambda: (P.update(V), P.sweeps(D.SWEEP_COUNT, l, E, lim, D.SWEEPS_TOLERANCE))),
                     ("potential ZERO scan", lambda: (P.update(V), P.sweeps(D.SWEEP_ZERO, l, E, None, D.SWEEPS_TOLERANCE))),
                     ("potential match", lambda: (P.update(V), P.match(l, E)))):
        fn(); t0 = time.time(); n = 30
        for _ in range(n): fn()
        print("%s %-50s %.3f ms per call" % (tag, name, (time.time() - t0) / n * 1e3))
    a = D.numerov_sweeps(ctx, grid, D.SWEEP_COUNT, V, [0, 1, 2, 3, 0], [-100.0, -50.0, -3.0, -1.0, -3000.0], [3, 3, 3, 3, 0])
    b = P.sweeps(D.SWEEP_COUNT, [0, 1, 2, 3, 0], [-100.0, -50.0, -3.0, -1.0, -3000.0], [3, 3, 3, 3, 0])
    c = P.sweeps(D.SWEEP_COUNT, [0, 1, 2, 3, 0], [-100.0, -50.0, -3.0, -1.0, -3000.0], [3, 3, 3, 3, 0], D.SWEEPS_TOLERANCE)
    za = D.numerov_sweeps(ctx, grid, D.SWEEP_ZERO, V, [0, 1, 2, 3], [-100.0, -50.0, -3.0, -1.0]); zb = P.sweeps(D.SWEEP_ZERO, [0, 1, 2, 3], [-100.0, -50.0, -3.0, -1.0])
    pa, ma = D.numerov_match(ctx, grid, V, [0, 2], [-100.0, -3.0]); pb, mb = P.match([0, 2], [-100.0, -3.0])
    print(tag, "equal:", np.array_equal(a["count"], b["count"]), np.array_equal(a["count"], c["count"]), np.array_equal(a["trip"], b["trip"]), np.array_equal(za["u0"], zb["u0"]), np.array_equal(pa, pb), np.array_equal(ma, mb))
    V2 = V * 1.0000001
    P.update(V2)
    print(tag, "after update equal:", np.array_equal(D.numerov_sweeps(ctx, grid, D.SWEEP_ZERO, V2, [0], [-100.0])["u0"], P.sweeps(D.SWEEP_ZERO, [0], [-100.0])["u0"]))
    P.close(); grid.close()
# the reference's unmodified L3 on the compat classes: Rn, 17 levels, a few steps, exact and tolerance sweeps
exe = os.path.join(os.path.dirname(__file__), "..", "oracle", "_ref", "ref_l3_cli")
if os.path.exists(exe):
    for env in ({}, {"DFTA_COMPAT_SWEEPS": "tolerance"}):
        t0 = time.time()
        p = subprocess.Popen([exe, "86", "17", "0.5", "50", "0.0001", "0"], stdout=subprocess.PIPE, text=True, env=dict(os.environ, **env))
        stamps = []
        for ln in p.stdout:
            if ln.startswith("Step:"):
                stamps.append(time.time())
                if len(stamps) >= 5: break
        p.kill()
        if len(stamps) >= 3: print("ref_l3_cli Rn @131073", env or "exact", ": %.2f s per SCF step" % ((stamps[-1] - stamps[1]) / (len(stamps) - 2)))
