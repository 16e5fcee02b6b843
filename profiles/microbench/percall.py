import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dftatom_amd as D
ctx = D.Context(0)
for L, d, R in ((14, 5e-4, 25.0), (17, 1e-4, 50.0)):
    g = D.Grid(ctx, L, d, R)
    rr = g.r(); V = np.zeros(g.N); V[1:] = -86.0 / rr[1:]
    for kind, name in ((D.SWEEP_COUNT, "CountNodes"), (D.SWEEP_ZERO, "SolutionInZero")):
        D.numerov_sweeps(ctx, g, kind, V, [0], [-0.3], [5])
        t = time.time(); n = 50
        for k in range(n): D.numerov_sweeps(ctx, g, kind, V, [0], [-0.3 - 1e-3 * k], [5])
        dt = (time.time() - t) / n
        print("N=%d one %s call (host potential in, one trial, result out): %.2f ms" % (g.N, name, 1e3 * dt))
    t = time.time(); n = 10
    for k in range(n): D.numerov_match(ctx, g, V, [0], [-0.3])
    print("N=%d one Match call: %.2f ms" % (g.N, 1e3 * (time.time() - t) / n))
    ps = D.Poisson(ctx, g, 1); rho = 86 * np.exp(-2 * rr) / np.pi
    ps.solve([86], rho); t = time.time(); ps.solve([86], rho); print("N=%d one SolvePoissonNonUniform call: %.1f ms" % (g.N, 1e3 * (time.time() - t)))
    ps.close(); g.close()
