"""A/B of level-search layouts in ONE process: the same atoms advanced K SCF steps under each variant of the environment given
on the command line (each argument "NAME=VALUE;NAME=VALUE", "" = defaults), e.g.

    PT_L=17 PT_ZS="86*256" PT_STEPS=9 PT_SKIP=4 python profiles/packed_ab.py "DFTA_DEBUG=LEVELS_NOPACK" "" "DFTA_DEBUG=LEVELS_PACK_DSMALL=4"

prints ms per step (level search, its sweep kernels, multigrid), rounds and issued trials per useful one averaged over steps PT_SKIP.., and
whether energies / eigenvalues / executed-sweep counts of every step equal the first variant's bit for bit.
PT_ZS: "86*256" (256 Rn atoms), "1-86" (a range), "2,10,18" (a list); PT_L: 12 / 14 / 17 multigrid levels; PT_LSDA=1.
Results quoted in DESIGN.md: profiles/r03_packed_rounds_ab.txt."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dftatom_amd as D
ctx = D.Context(0)
L = int(os.environ.get("PT_L", "17"))
delta, R = {12: (2e-3, 25.0), 14: (5e-4, 25.0), 17: (1e-4, 50.0)}[L]
grid = D.Grid(ctx, L, delta, R)
spec = os.environ.get("PT_ZS", "86*16")
if "*" in spec:
    z, n = spec.split("*"); Zs = [int(z)] * int(n)
elif "-" in spec:
    a, b = spec.split("-"); Zs = list(range(int(a), int(b) + 1))
else:
    Zs = [int(z) for z in spec.split(",")]
steps = int(os.environ.get("PT_STEPS", "12"))
skip = int(os.environ.get("PT_SKIP", "4"))
lsda = bool(int(os.environ.get("PT_LSDA", "0")))
res = {}
for var in sys.argv[1:] or [""]:
    envs = dict(kv.split("=", 1) for kv in var.split(";") if kv)
    old = {k: os.environ.get(k) for k in envs}
    os.environ.update(envs)
    scf = D.Scf(ctx, grid, Zs, lsda=lsda)
    tr = []
    agg = dict(ms=0.0, lv=0.0, po=0.0, sw=0.0, rounds=0, issued=0, ref=0, n=0)
    for it in range(steps):
        t0 = time.time()
        st = scf.step()
        dt = (time.time() - t0) * 1e3
        es, fin = scf.energies()
        lv = [float(e) for a in range(len(Zs)) for sp in range(2 if lsda else 1) for e in scf.levels(a, sp)["E"]] + [e.Ekinetic for e in es] + [int(st.sweeps_reference_executed)]
        tr.append(([e.Etotal for e in es], lv))
        if it >= skip:
            agg["ms"] += dt; agg["lv"] += st.ms_levels; agg["po"] += st.ms_poisson; agg["sw"] += st.ms_sweep_kernels; agg["rounds"] += st.rounds
            agg["issued"] += st.sweeps_issued; agg["ref"] += st.sweeps_reference_executed; agg["n"] += 1
    n = max(agg["n"], 1)
    print("%-44s %d atoms: %.1f ms/step (levels %.1f [sweeps %.1f], poisson %.1f) rounds %.1f issued/useful %.2f  info %s"
          % (var or "(default)", len(Zs), agg["ms"] / n, agg["lv"] / n, agg["sw"] / n, agg["po"] / n, agg["rounds"] / n, agg["issued"] / max(agg["ref"], 1), (scf.tree_depth, scf.njobs, scf.trials_per_round)), flush=True)
    scf.close()
    for k, v in old.items():
        if v is None: os.environ.pop(k)
        else: os.environ[k] = v
    res[var] = tr
vs = list(res)
for v in vs[1:]:
    bad = 0
    for it, (x, y) in enumerate(zip(res[vs[0]], res[v])):
        if x[0] != y[0] or x[1] != y[1]:
            bad += 1
            if bad <= 3:
                de = max(abs(p - q) for p, q in zip(x[0], y[0]))
                nl = sum(1 for p, q in zip(x[1], y[1]) if p != q)
                print("   step %d differs from first variant: max |dEtotal| %.3e, %d level records differ" % (it, de, nl))
    print("%-44s %s" % (v, "bit-identical to first variant over %d steps" % steps if not bad else "DIFFERS in %d steps" % bad))
