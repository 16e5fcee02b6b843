"""time one Poisson solve (Rn-like density, L levels) for the env variants given on the command line; compare U bits"""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dftatom_amd as D
L = int(os.environ.get("PT_L", "17"))
B = int(os.environ.get("PT_B", "1"))
delta, R = {14: (5e-4, 25.0), 17: (1e-4, 50.0), 20: (1.25e-5, 50.0), 12: (2e-3, 25.0)}[L]
ctx = D.Context(0)
grid = D.Grid(ctx, L, delta, R)
rr = grid.r()
Z = int(os.environ.get("PT_Z", "86"))
rho = np.tile(Z * np.exp(-2 * rr) / np.pi, (B, 1))
ref = None
for var in sys.argv[1:] or [""]:
    envs = dict(kv.split("=", 1) for kv in var.split(";") if kv)
    old = {k: os.environ.get(k) for k in envs}
    os.environ.update(envs)
    ps = D.Poisson(ctx, grid, B)
    U, vc, err = ps.solve([Z] * B, rho)
    ts = []
    for _ in range(5):
        ps.solve([Z] * B, rho)
        ts.append(ctx.last_kernel_ms())
    G = ps.group_info()
    ps.close()
    for k, v in old.items():
        if v is None: os.environ.pop(k)
        else: os.environ[k] = v
    same = "" if ref is None else (" bit-identical to first" if np.array_equal(U.view(np.int64), ref.view(np.int64)) else " DIFFERS from first: max |dU| %.3e" % np.max(np.abs(U - ref)))
    if ref is None: ref = U.copy()
    print("%-40s G=%s vcycles %d err %.17g  kernel ms: min %.3f med %.3f%s" % (var or "(default)", G, vc[0], err[0], min(ts), sorted(ts)[2], same), flush=True)
