"""A/B of the batch level search (round 6, DESIGN_LOG.md r6 items 1 / 1b): level phase per SCF step (HIP events inside the library) of
Z = 1..86 (`pt`), a 12-atom shard (`shard`), 16 / 64 / 256 Rn (`b16`, `b64`, `b256`) and 16 x Rn LSDA @ 2^20+1 (`l20`) under DFTA_DEBUG variants:

    VARIANTS="|LEVELS_NOQUEUE|LEVELS_OWN" python profiles/batch_levels_ab.py pt shard b64 b256 l20

("" = the default: packed / static host rounds, sweep blocks launched longest first).  Every variant must print the same E0."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import dftatom_amd as D
from golden.make_golden import GRIDS

def run(ctx, grid, Z, lsda, warm, steps, knobs):
    if knobs: os.environ["DFTA_DEBUG"] = knobs
    else: os.environ.pop("DFTA_DEBUG", None)
    scf = D.Scf(ctx, grid, Z, lsda=lsda)
    os.environ.pop("DFTA_DEBUG", None)
    for _ in range(warm): scf.step()
    tl = tp = tk = 0.0; rounds = 0; lay = set()
    t0 = time.time()
    for _ in range(steps):
        st = scf.step()
        tl += st.ms_levels; tp += st.ms_poisson; tk += st.ms_sweep_kernels; rounds += st.rounds; lay.add(int(st.levels_layout))
    wall = (time.time() - t0) / steps * 1e3
    e = scf.energies()[0][0].as_list()[0]
    scf.close()
    return {"levels_ms": round(tl / steps, 2), "sweep_kernels_ms": round(tk / steps, 2), "poisson_ms": round(tp / steps, 2), "step_ms": round(wall, 2), "rounds": rounds / steps, "layout": sorted(lay), "E0": e}

ctx = D.Context(0)
which = sys.argv[1:] or ["pt", "shard", "l20", "b64", "b256"]
variants = os.environ.get("VARIANTS", "|LEVELS_NOOWN").split("|")
g17 = D.Grid(ctx, *GRIDS["L17"])
cases = {"pt": (g17, list(range(1, 87)), False, 3, 4), "shard": (g17, list(range(75, 87)), False, 4, 5), "shard2": (g17, [3, 11, 19, 30, 37, 48, 55, 62, 70, 79, 86], False, 4, 5),
         "b64": (g17, [86] * 64, False, 3, 4), "b256": (g17, [86] * 256, False, 3, 4), "b16": (g17, [86] * 16, False, 4, 5)}
for name in which:
    if name == "l20":
        g = D.Grid(ctx, *GRIDS["L20"]); case = (g, [86] * 16, True, 2, 3)
    elif name.startswith("z:"):                    # "z:77-86": Z = 77 .. 86 at 131 073 nodes
        lo, hi = (int(x) for x in name[2:].split("-"))
        case = (g17, list(range(lo, hi + 1)), False, 4, 5)
    else:
        case = cases[name]
    for kn in variants:
        r = run(ctx, case[0], case[1], case[2], case[3], case[4], kn)
        print(name, repr(kn), json.dumps(r), flush=True)
    if name == "l20": g.close()
