"""How predictable is the movement of a level's eigenvalue from one SCF step to the next?  (The evidence behind Job::hist_c / hist_w,
csrc/levels.hip: the bracket from which the first spine of every bisection is planned.)

Runs eight atoms for up to 40 SCF steps on the exact kernels, records every level's eigenvalue per step, and replays two rules on the
series: the two-step rule of rounds 3-4 (next end point within T +- 2 |d|, d = last movement) and the extrapolation of round 5
(T + q d +- (a + b |q - q_prev|) |d|, q = d / d_prev, used when |q|, |q_prev| <= 1).  Per atom: bits gained (log2 of the ratio of the two
bracket widths, floors 1e-10 |T| + 64e-12 included), how often each bracket would have missed the next end point (a miss costs the round
its tree, never a result), and the ratio q itself.  Speculation only -- nothing here can change a result.

    python3 profiles/history_brackets.py --out profiles/r05_history_brackets.json        (on the MI355X box)
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K_ERR = 1e-12


def floor_of(T):
    return 1e-10 * abs(T) + 64 * K_ERR


def replay(E, a, b, first=3, last=25):
    E = np.array(E)
    n, L = E.shape
    gain, miss_new, miss_old, cases, qs = [], 0, 0, 0, []
    for k in range(first, min(n - 1, last)):
        for j in range(L):
            T, d, d1, d2 = E[k, j], E[k, j] - E[k - 1, j], E[k - 1, j] - E[k - 2, j], E[k - 2, j] - E[k - 3, j]
            w_old, c, w = 2 * abs(d) + floor_of(T), T, None
            w = w_old
            if d1 != 0 and d2 != 0:
                q, q0 = d / d1, d1 / d2
                qs.append(q)
                if abs(q) <= 1.0 and abs(q0) <= 1.0:
                    u = (a + b * abs(q - q0)) * abs(d)
                    if u < 2 * abs(d):
                        c, w = T + q * d, u + floor_of(T + q * d)
            nxt = E[k + 1, j]
            cases += 1
            miss_new += abs(nxt - c) > w
            miss_old += abs(nxt - T) > w_old
            gain.append(float(np.log2(w_old / w)))
    return {"cases": cases, "bits_gained_mean": float(np.mean(gain)), "misses_extrapolated": int(miss_new), "misses_two_step": int(miss_old),
            "q_median": float(np.median(qs)) if qs else None, "q_p10": float(np.percentile(qs, 10)) if qs else None, "q_p90": float(np.percentile(qs, 90)) if qs else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_history_brackets.json"))
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    import dftatom_amd as D
    ctx = D.Context(0)
    grid = D.Grid(ctx, 17, 1e-4, 50.0)
    out = {"workload": "eigenvalue of every level per SCF step, 131 073 nodes, exact kernels; rules replayed on steps 4 .. 25 (the bench times steps 5 .. 24)",
           "rule_in_the_library": "T + q d +- (0.2 + 4 |q - q_prev|) |d|", "atoms": {}}
    for Z, lsda in ((86, False), (86, True), (29, True), (64, False), (18, False), (47, False), (26, True), (55, False)):
        scf = D.Scf(ctx, grid, [Z], lsda=lsda)
        Es = []
        for _ in range(a.steps):
            scf.step(want_stats=False)
            Es.append([float(x) for x in scf.levels(0, 0)["E"]])
            if scf.energies()[1].all():
                break
        scf.close()
        out["atoms"]["Z=%d %s" % (Z, "LSDA (spin 0)" if lsda else "LDA")] = {"steps": len(Es), "levels": len(Es[0]),
                                                                              "a=0.2 b=4 (library)": replay(Es, 0.2, 4.0), "a=0.1 b=4": replay(Es, 0.1, 4.0), "a=0.05 b=3": replay(Es, 0.05, 3.0)}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    for k, v in out["atoms"].items():
        print(k, v["a=0.2 b=4 (library)"])


if __name__ == "__main__":
    main()
