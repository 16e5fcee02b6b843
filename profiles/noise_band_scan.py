# width of the energy band in which (a) CountNodes and (b) the sign of u(0) are not monotonic, per level of Rn (potential after 12 SCF steps)
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dftatom_amd as D
ctx = D.Context(0)
g = D.Grid(ctx, 17, 1e-4, 50.0)
scf = D.Scf(ctx, g, [86])
for _ in range(12): scf.step()
V = scf.array(3)
lev = scf.levels(0, 0)
def count(l, Es):
    Es = np.asarray(Es, dtype=np.float64)
    return D.numerov_sweeps(ctx, g, D.SWEEP_COUNT, V, np.full(Es.size, l, np.int32), Es, np.full(Es.size, 1000, np.int32))["count"]
def u0pos(l, Es):
    Es = np.asarray(Es, dtype=np.float64)
    return D.numerov_sweeps(ctx, g, D.SWEEP_ZERO, V, np.full(Es.size, l, np.int32), Es)["u0"] > 0
def mixed(pred, c0, label):
    out = []
    for rel in (1e-9, 1e-10, 1e-11, 1e-12):
        w = rel * abs(c0)
        Es = c0 + np.linspace(-w, w, 16384)
        a = pred(Es)
        if a[0] == a[-1]: out.append("%.0e: no change" % rel); break
        up = a != a[0]
        first = int(np.argmax(up)); last = len(Es) - 1 - int(np.argmax(~up[::-1]))
        if last > first:
            out.append("%.0e: mixed %.2e|E| (%.1e abs, %d flips)" % (rel, (Es[last] - Es[first]) / abs(c0), Es[last] - Es[first], int(np.sum(a[1:] != a[:-1]))))
            c0 = 0.5 * (Es[last] + Es[first])
        else:
            out.append("%.0e: clean" % rel)
            c0 = 0.5 * (Es[first - 1] + Es[first])
    print(label, " | ".join(out), flush=True)
def locate(pred, lo, hi):
    a = pred([lo])[0]
    for it in range(100):
        mid = 0.5 * (lo + hi)
        if pred([mid])[0] != a: hi = mid
        else: lo = mid
        if hi - lo < 2e-10 * abs(lo): break
    return 0.5 * (lo + hi)
for k in range(len(lev['n'])):
    n, l, E = int(lev['n'][k]), int(lev['l'][k]), float(lev['E'][k])
    # (b) sign of u(0) around the eigenvalue
    lo, hi = E - 2e-3 * abs(E), E + 2e-3 * abs(E)
    if u0pos(l, [lo])[0] != u0pos(l, [hi])[0]:
        mixed(lambda Es: u0pos(l, Es), locate(lambda Es: u0pos(l, Es), lo, hi), "lev %2d l%d E %12.6f u(0) sign :" % (k, l, E))
    else:
        print("lev %2d l%d E %12.6f u(0) sign : no change within 2e-3" % (k, l, E))
    # (a) first change of the count above the eigenvalue (top of the band)
    lo = E - 2e-3 * abs(E)
    c_lo = count(l, [lo])[0]
    hi = None
    for rel in (2e-3, 1e-2, 0.1, 0.3, 0.6, 0.9, 0.99):
        t = E + rel * abs(E)
        if count(l, [t])[0] != c_lo: hi = t; break
    if hi is None: print("lev %2d count: no change below 0" % k); continue
    mixed(lambda Es: count(l, Es), locate(lambda Es: count(l, Es), lo, hi), "lev %2d l%d E %12.6f count top  :" % (k, l, E))
