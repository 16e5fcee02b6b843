#!/usr/bin/env python3
"""Generate tests/golden/* from the COMPILED REFERENCE (oracle/_ref/libdfta_ref.so).

Run in the development container only (needs /root/reference to build oracle/_ref):
    make -C oracle ref && python tests/golden/make_golden.py [--rn]

Every value stored here is an output of the reference's own code (DFT::Numerov, DFT::PoissonSolver,
DFT::VWNExchCor, DFT::Integral, DFT::AufbauPrinciple, DFT::DFTAtom) called through
oracle/ref_harness.cpp; inputs are generated deterministically in this script.  Fixtures are data
(inputs + expected outputs), no reference source text.

--rn additionally runs the reference end to end for Radon (Z=86, 17 levels, 131073 nodes, LDA and
LSDA; about 10 minutes) and stores the final-step values in rn_end_to_end.json.
"""
import ctypes as C
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402

GRIDS = {"L14": (14, 5e-4, 25.0), "L17": (17, 1e-4, 50.0), "L20": (20, 1.25e-5, 50.0)}


def screened_potential(r, Z):
    """Deterministic Thomas-Fermi-like screened potential used as a stand-in SCF potential."""
    V = np.zeros_like(r)
    x = r[1:]
    V[1:] = -(1.0 + (Z - 1.0) * np.exp(-0.9 * Z ** (1.0 / 3.0) * x)) / x
    return V


def ref_text(r, mode, Z, L, alpha, R, d, hp=True):
    buf = C.create_string_buffer(1 << 23)
    (r.ref_calculate_hp if hp else r.ref_calculate)(mode, Z, L, alpha, R, d, buf, len(buf))
    return buf.value.decode()


def parse_run(txt):
    """-> list of steps: {levels: [(label, E, nodes)], energies: {...}}"""
    steps = []
    cur = None
    for ln in txt.splitlines():
        if ln.startswith("Step:"):
            cur = {"levels": [], "energies": None}
            steps.append(cur)
        elif ln.startswith("Energy") and cur is not None:
            m = re.match(r"Energy (?:alpha |beta )?(\d+\w): (\S+) Num nodes: (\d+)", ln)
            cur["levels"].append([m.group(1), float(m.group(2)), int(m.group(3))])
        elif ln.startswith("Etotal") and cur is not None:
            m = re.match(r"Etotal = (\S+) Ekin = (\S+) Ecoul = (\S+) Eenuc = (\S+) Exc = (\S+)", ln)
            cur["energies"] = [float(m.group(k)) for k in range(1, 6)]
    return steps


def main():
    r = O.ref()
    o = O.oracle()
    out = {}
    meta = {}
    rng = np.random.default_rng(20260206)

    # 1. grids -------------------------------------------------------------------------------
    for name, (L, d, R) in GRIDS.items():
        N = r.ref_num_nodes(L)
        V0 = np.zeros(N)
        h = r.ref_numerov_create(O.dp(V0), N, d, R)
        idx = np.unique(np.concatenate([np.arange(0, 8), np.linspace(8, N - 1, 40).astype(np.int64)]))
        # r_i through GetBoundaryValueZero: pow(r_i, 1) * exp(-i*d/2) is not r_i; use f() instead:
        # with V=0, l=0: f(i) = 2(0-E) Rp2delta2 exp(2 i d) + d^2/4 -> store f for E=-1 as the grid golden
        fvals = np.array([r.ref_f(h, 0, -1.0, int(i)) for i in idx])
        f3 = np.array([r.ref_f(h, 3, -2.5, int(i)) for i in idx[1:]])
        out[f"grid_{name}_idx"] = idx
        out[f"grid_{name}_f_l0_Em1"] = fvals
        out[f"grid_{name}_f_l3_Em2p5"] = f3
        meta[f"grid_{name}"] = {"L": L, "delta": d, "Rmax": R, "N": N, "Rp": r.ref_rp(h)}
        r.ref_numerov_destroy(h)

    # 2. Numerov kernel tables ---------------------------------------------------------------
    L, d, R = GRIDS["L14"]
    g = O.make_grid(L, d, R)
    N = g.N
    rr = O.grid_r(g)
    pots = {"coulomb18": O.coulomb_potential(g, 18), "screened18": screened_potential(rr, 18.0),
            "screened86": screened_potential(rr, 86.0)}
    for pname, V in pots.items():
        h = r.ref_numerov_create(O.dp(V), N, d, R)
        Zp = 86.0 if pname.endswith("86") else 18.0
        rows = []
        psis = []
        Es = np.concatenate([-rng.uniform(1e-3, Zp * Zp + 1, 24), -10.0 ** rng.uniform(-3, 1, 8),
                             [-(Zp ** 2) / 2, -(Zp ** 2) / 8, -(Zp ** 2) / 18, 0.5, 25.0, 50.0]])
        for l in range(4):
            for E in Es:
                for lim in (0, 2, 5):
                    cnt = r.ref_count_nodes(h, l, float(E), lim)
                    rows.append([l, E, lim, cnt])
                u0 = r.ref_solution_in_zero(h, l, float(E))
                cut = r.ref_max_radius_index(h, float(E))
                P = np.zeros(N)
                mp = r.ref_match(h, l, float(E), O.dp(P))
                samp = P[:: max(1, N // 64)].copy()
                psis.append(np.concatenate([[l, E, u0, cut, mp, np.nansum(P), np.nansum(np.abs(P))], samp]))
        out[f"numerov_{pname}_counts"] = np.array(rows)
        out[f"numerov_{pname}_sweeps"] = np.array(psis)
        r.ref_numerov_destroy(h)
    out["pot_screened18_L14"] = pots["screened18"]

    # 3. level driver --------------------------------------------------------------------------
    for pname, Z in (("screened18", 18), ("screened86", 86)):
        V = pots[pname]
        lv = O.subshells(Z)
        h = r.ref_numerov_create(O.dp(V), N, d, R)
        n = np.array([a for a, _, _ in lv], np.int32)
        l = np.array([b for _, b, _ in lv], np.int32)
        occ = np.array([c for _, _, c in lv], np.int32)
        E = np.zeros(len(lv))
        nd = np.zeros(N)
        Eel = C.c_double(0)
        Bot = C.c_double(-float(Z) * Z - 1.0)
        conv = r.ref_loop_over_levels(h, len(lv), O.ip(n), O.ip(l), O.ip(occ), O.dp(E), O.dp(nd), C.byref(Eel),
                                      C.byref(Bot), d)
        out[f"levels_{pname}_E"] = E
        out[f"levels_{pname}_newdensity_sample"] = nd[:: N // 256].copy()
        out[f"levels_{pname}_scalars"] = np.array([Eel.value, Bot.value, conv, nd.sum()])
        # LocateInterval alone, first three levels, fresh bottom
        li = []
        for k in range(min(4, len(lv))):
            top = C.c_double(50.0)
            bot = C.c_double(-float(Z) * Z - 1.0)
            r.ref_locate_interval(h, C.byref(top), C.byref(bot), int(l[k]), int(n[k] - l[k]))
            li.append([n[k], l[k], top.value, bot.value])
        out[f"levels_{pname}_locate"] = np.array(li)
        r.ref_numerov_destroy(h)

    # 4. Poisson ---------------------------------------------------------------------------------
    Lp, dp_, Rp_ = 12, 2e-3, 25.0
    gp = O.make_grid(Lp, dp_, Rp_)
    rp = O.grid_r(gp)
    for tag, Z in (("H", 1), ("Z18", 18), ("Z86", 86)):
        rho = Z * np.exp(-2 * rp) / np.pi
        q = r.ref_poisson_create(Lp, dp_)
        U = np.zeros(gp.N)
        r.ref_solve_poisson_nonuniform(q, Z, Rp_, O.dp(rho), gp.N, O.dp(U))
        out[f"poisson_{tag}_U"] = U
        r.ref_poisson_destroy(q)
    meta["poisson_grid"] = {"L": Lp, "delta": dp_, "Rmax": Rp_}
    # single GS / restrict / prolong on small levels
    Ls, ds = 8, 0.01
    q = r.ref_poisson_create(Ls, ds)
    phis, srcs = [], []
    for lvl in range(Ls):
        nl = r.ref_poisson_level_size(q, lvl)
        phi = rng.standard_normal(nl)
        src = rng.standard_normal(nl)
        r.ref_poisson_set_level(q, lvl, O.dp(phi), O.dp(src))
        phis.append(phi)
        srcs.append(src)
        out[f"mg_in_phi_{lvl}"] = phi
        out[f"mg_in_src_{lvl}"] = src

    def snap(tag):
        for lvl in range(Ls):
            nl = r.ref_poisson_level_size(q, lvl)
            a = np.zeros(nl)
            b = np.zeros(nl)
            r.ref_poisson_get_level(q, lvl, O.dp(a), O.dp(b))
            out[f"mg_{tag}_phi_{lvl}"] = a
            out[f"mg_{tag}_src_{lvl}"] = b
    errs = [r.ref_gauss_seidel(q, lvl) for lvl in range(Ls)]
    out["mg_gs_err"] = np.array(errs)
    snap("gs")
    for lvl in range(1, Ls):
        r.ref_restrict(q, lvl)
    snap("restrict")
    for lvl in range(Ls - 1, 0, -1):
        r.ref_prolong(q, lvl)
    snap("prolong")
    out["mg_vcycle_err"] = np.array([r.ref_vcycle(q, 1e-14, 3)])
    snap("vcycle")
    meta["mg_small"] = {"L": Ls, "delta": ds}
    r.ref_poisson_destroy(q)

    # 5. VWN ---------------------------------------------------------------------------------------
    n = np.concatenate([10.0 ** np.linspace(-20, 6, 261), [0.0, 9.99e-19, 1e-18]])
    a = np.zeros_like(n)
    r.ref_vwn_vexc(O.dp(n), O.dp(a), len(n))
    out["vwn_n"] = n
    out["vwn_vexc"] = a.copy()
    r.ref_vwn_eexcdif(O.dp(n), O.dp(a), len(n))
    out["vwn_eexcdif"] = a.copy()
    for zeta in (0.0, 0.3, -0.3, 1.0, -1.0, 0.77):
        na = n * (1 + zeta) / 2
        nb = n * (1 - zeta) / 2
        res, va, vb = np.zeros_like(n), np.zeros_like(n), np.zeros_like(n)
        r.ref_vwn_vexc_lsda(O.dp(na), O.dp(nb), O.dp(res), O.dp(va), O.dp(vb), len(n))
        e = np.zeros_like(n)
        r.ref_vwn_eexcdif_lsda(O.dp(na), O.dp(nb), O.dp(e), len(n))
        out[f"vwn_lsda_z{zeta}"] = np.stack([na, nb, res, va, vb, e])

    # 6. integrals -----------------------------------------------------------------------------------
    gi = O.make_grid(10, 8e-3, 25.0)
    ri = O.grid_r(gi)
    i_ = np.arange(gi.N)
    integrand = 4 * np.pi * ri ** 2 * (np.exp(-2 * ri) / np.pi) * (gi.Rp * gi.delta * np.exp(gi.delta * i_))
    out["int_integrand"] = integrand
    out["int_values"] = np.array([r.ref_integral(k, 1.0, O.dp(integrand), gi.N) for k in range(5)])
    noise = rng.standard_normal(4097)
    out["int_noise"] = noise
    out["int_noise_values"] = np.array([r.ref_integral(k, 0.37, O.dp(noise), 4097) for k in range(5)])

    # 7. Aufbau ----------------------------------------------------------------------------------------
    auf = {}
    for Z in range(1, 119):
        an, al, ao = (np.zeros(32, np.int32) for _ in range(3))
        c = r.ref_get_subshells(Z, O.ip(an), O.ip(al), O.ip(ao))
        auf[str(Z)] = [[int(an[i]), int(al[i]), int(ao[i])] for i in range(c)]
    meta["aufbau"] = auf

    # 8. end to end, Ar 14 levels (README.md:62-76 configuration) ------------------------------------------
    e2e = {}
    for mode, tag in ((0, "Ar_LDA_L14"), (1, "Ar_LSDA_L14")):
        txt = ref_text(r, mode, 18, 14, 0.5, 25.0, 5e-4, hp=True)
        steps = parse_run(txt)
        e2e[tag] = {"nsteps": len(steps), "finished": "Finished!" in txt,
                    "steps": [s for s in steps[:3]] + [steps[-1]],
                    "etotal_all": [s["energies"][0] for s in steps],
                    "config_line": txt.strip().splitlines()[-1 if mode == 0 else -2:]}
        if mode == 0:
            txt6 = ref_text(r, 0, 18, 14, 0.5, 25.0, 5e-4, hp=False)
            e2e["Ar_LDA_L14_text_tail"] = txt6.strip().splitlines()[-10:]
    meta["end_to_end"] = e2e

    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
    with open(os.path.join(HERE, "golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote golden.npz (%d arrays), golden_meta.json" % len(out))

    if "--rn" in sys.argv:
        rn = {}
        for mode, tag in ((0, "Rn_LDA_L17"), (1, "Rn_LSDA_L17")):
            txt = ref_text(r, mode, 86, 17, 0.5, 50.0, 1e-4, hp=True)
            steps = parse_run(txt)
            rn[tag] = {"nsteps": len(steps), "finished": "Finished!" in txt, "first": steps[0], "second": steps[1],
                       "last": steps[-1], "etotal_all": [s["energies"][0] for s in steps]}
            with open(os.path.join(HERE, "rn_end_to_end.json"), "w") as f:
                json.dump(rn, f, indent=1)
            print("done", tag, steps[-1]["energies"])


if __name__ == "__main__":
    main()
