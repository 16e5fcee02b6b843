#!/usr/bin/env python3
"""Golden values for the BASELINE configs that round 1 left without fixtures, from the COMPILED REFERENCE
(oracle/_ref/libdfta_ref.so, built by `make -C oracle ref` from /root/reference where it lies).

    python tests/golden/make_golden_table.py table [--jobs 6]    -> periodic_table_L17.json  (config 4)
    python tests/golden/make_golden_table.py l20                 -> l20.npz + l20_meta.json   (config 5)
    python tests/golden/make_golden_table.py uniform             -> uniform.npz + uniform_meta.json
    python tests/golden/make_golden_table.py l20cond             -> l20_meta.json["poisson_conditioning"] (reference's max |dU| for a 1e-12 density perturbation)
    python tests/golden/make_golden_table.py jitter [--jobs 4]   -> late_step_jitter.json (eigenvalues of the reference's last steps: Rn and the worst atoms of the table)

table: DFT::DFTAtom::CalculateNonUniformLDA(Z, 17, 0.5, 50, 1e-4) for Z = 1..118 (OptionsFrame.cpp:153 allows Z <= 118; round 3
       added 87..118: Ac/Th/Pa/U/Np/Cm/Lr exceptions of AufbauPrinciple.h:101-117), run to the reference's own stop
       (17-digit console protocol of oracle/ref_hp.cpp); stored per atom: number of steps, Finished flag, the
       first two steps and the last one (eigenvalues + the five printed energies).  About 3 CPU-hours, spread
       over a process pool.
l20  : 1 048 577-node grid (20 levels, delta = 1.25e-5, Rmax = 50): CountNodes / u(0) / matchPoint rows on a screened
       Rn-like potential, one SolvePoissonNonUniform, and the first two SCF steps of Rn LSDA (the capture buffer stops
       the run when "Step: 2" is printed).

Fixtures are data (inputs are regenerated deterministically by the tests; outputs are stored here).
"""
import ctypes as C
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402
from make_golden import parse_run, screened_potential  # noqa: E402


def run_ref(mode, Z, L, alpha, R, d, max_steps=-1):
    r = O.ref()
    buf = C.create_string_buffer(1 << 23)
    r.ref_calculate_hp_steps(mode, Z, L, alpha, R, d, max_steps, buf, len(buf))
    return buf.value.decode()


def one_atom(Z):
    t0 = time.time()
    txt = run_ref(0, Z, 17, 0.5, 50.0, 1e-4)
    steps = parse_run(txt)
    rec = {"Z": Z, "nsteps": len(steps), "finished": "Finished!" in txt, "first": steps[0], "second": steps[1],
           "last": steps[-1], "etotal_all": [s["energies"][0] for s in steps],
           "config": txt.strip().splitlines()[-1], "seconds": round(time.time() - t0, 1)}
    return rec


def table(jobs, zmax=118):
    path = os.path.join(HERE, "periodic_table_L17.json")
    done = {}
    if os.path.exists(path):
        with open(path) as f:
            done = json.load(f)
    todo = [Z for Z in range(zmax, 0, -1) if str(Z) not in done]      # heavy atoms first: better packing
    with mp.Pool(jobs) as pool:
        for rec in pool.imap_unordered(one_atom, todo):
            done[str(rec["Z"])] = rec
            with open(path + ".tmp", "w") as f:
                json.dump(done, f, indent=0, sort_keys=True)
            os.replace(path + ".tmp", path)
            print("Z=%d: %d steps, finished=%s, Etotal=%.9f (%.0f s)" % (rec["Z"], rec["nsteps"], rec["finished"],
                                                                       rec["last"]["energies"][0], rec["seconds"]), flush=True)


def l20():
    r = O.ref()
    L, d, R = 20, 1.25e-5, 50.0
    g = O.make_grid(L, d, R)
    N = g.N
    rr = O.grid_r(g)
    V = screened_potential(rr, 86.0)
    out, meta = {}, {"grid": {"L": L, "delta": d, "Rmax": R, "N": N}}
    rng = np.random.default_rng(20261002)
    h = r.ref_numerov_create(O.dp(V), N, d, R)
    rows = []
    Es = np.concatenate([-rng.uniform(1e-2, 86.0 ** 2 + 1, 10), [-3204.75, -0.29, -86.0 ** 2 / 2, 0.5, 50.0]])
    for l in range(4):
        for E in Es:
            cnt = r.ref_count_nodes(h, l, float(E), 3)
            u0 = r.ref_solution_in_zero(h, l, float(E))
            cut = r.ref_max_radius_index(h, float(E))
            rows.append([l, E, 3, cnt, u0, cut])
    out["sweeps"] = np.array(rows)
    P = np.zeros(N)
    ms = []
    for l, E in ((0, -3000.0), (1, -500.0), (3, -8.0)):
        mp_ = r.ref_match(h, l, E, O.dp(P))
        ms.append(np.concatenate([[l, E, mp_, np.nansum(P), np.nansum(np.abs(P))], P[:: N // 64][:64]]))
    out["match"] = np.array(ms)
    r.ref_numerov_destroy(h)
    print("sweeps done", flush=True)
    # one Poisson solve: Rn-like density 86 * exp(-2 r) / pi scaled to 86 electrons
    rho = 86.0 * np.exp(-2 * rr) / np.pi
    q = r.ref_poisson_create(L, d)
    U = np.zeros(N)
    t0 = time.time()
    r.ref_solve_poisson_nonuniform(q, 86, R, O.dp(rho), N, O.dp(U))
    r.ref_poisson_destroy(q)
    out["poisson_U_sample"] = U[:: 64].copy()          # every 64th node: 16385 values
    out["poisson_U_checksum"] = np.array([U.sum(), np.abs(U).sum(), U[1], U[N // 2], U[N - 2]])
    print("poisson done %.0f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, "l20.npz"), **out)
    t0 = time.time()
    txt = run_ref(1, 86, L, 0.5, R, d, max_steps=2)
    steps = parse_run(txt)
    meta["Rn_LSDA_L20"] = {"steps": steps[:2]}
    print("scf done %.0f s" % (time.time() - t0), flush=True)
    with open(os.path.join(HERE, "l20_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


def l20cond():
    """Conditioning of SolvePoissonNonUniform at 1 048 577 nodes, measured ON THE REFERENCE: the same density with and without a seeded
    relative perturbation of 1e-12 -> max |dU| (the 100-V-cycle end state is a round-off floor: second differences over 2^20 nodes cancel
    ten digits).  tests/test_gpu_configs.py gates SCF step 1 at this size with these numbers (not with the product's own)."""
    r = O.ref()
    L, d, R = 20, 1.25e-5, 50.0
    g = O.make_grid(L, d, R)
    N = g.N
    rr = O.grid_r(g)
    rho = 86.0 * np.exp(-2 * rr) / np.pi
    rng = np.random.default_rng(5)
    rho2 = rho * (1.0 + 1e-12 * rng.standard_normal(N))
    U = [np.zeros(N), np.zeros(N)]
    for k, dens in enumerate((rho, rho2)):
        q = r.ref_poisson_create(L, d)
        t0 = time.time()
        r.ref_solve_poisson_nonuniform(q, 86, R, O.dp(np.ascontiguousarray(dens)), N, O.dp(U[k]))
        r.ref_poisson_destroy(q)
        print("solve %d: %.0f s" % (k, time.time() - t0), flush=True)
    kabs = float(np.max(np.abs(U[0] - U[1])))
    krel = float(np.max(np.abs(U[0][1:] - U[1][1:]) / np.abs(U[0][1:])))
    path = os.path.join(HERE, "l20_meta.json")
    with open(path) as f:
        meta = json.load(f)
    meta["poisson_conditioning"] = {"density": "86 exp(-2 r) / pi", "perturbation": "rho * (1 + 1e-12 * default_rng(5).standard_normal(N))",
                                    "max_abs_dU": kabs, "max_rel_dU": krel}
    with open(path, "w") as f:
        json.dump(meta, f, indent=1)
    print("reference: max |dU| = %.3e, relative %.3e" % (kabs, krel))


def _jitter_atom(Z):
    t0 = time.time()
    txt = run_ref(0, Z, 17, 0.5, 50.0, 1e-4)
    steps = parse_run(txt)
    return {"Z": Z, "nsteps": len(steps), "finished": "Finished!" in txt, "last_steps": steps[-4:], "seconds": round(time.time() - t0, 1)}


def jitter(jobs, atoms=(86, 111, 118, 28, 46, 52, 64, 79)):
    """The eigenvalues of the compiled reference's LAST FOUR SCF steps (L = 17) for Rn and a sample of the atoms whose final states the GPU
    tests gate loosest: how far the reference's own eigenvalues still move from step to step when it stops (or hits its 100-step cap) is
    what bounds any comparison of "converged" eigenvalues (tests/test_oracle_golden.py::test_reference_late_step_jitter)."""
    path = os.path.join(HERE, "late_step_jitter.json")
    out = {}
    with mp.Pool(jobs) as pool:
        for rec in pool.imap_unordered(_jitter_atom, list(atoms)):
            out[str(rec["Z"])] = rec
            print("Z=%d: %d steps, finished=%s (%.0f s)" % (rec["Z"], rec["nsteps"], rec["finished"], rec["seconds"]), flush=True)
            with open(path, "w") as f:
                json.dump(out, f, indent=0, sort_keys=True)


def uniform():
    """Uniform-grid path (NumerovFunctionRegularGrid, SolvePoissonUniform, CalculateUniformLDA/LSDA): -> uniform.npz / uniform_meta.json"""
    r = O.ref()
    out, meta = {}, {}
    rng = np.random.default_rng(20261003)
    L, R = 14, 25.0
    N = r.ref_num_nodes(L)
    h = R / (N - 1)
    rr = h * np.arange(N)
    meta["grid"] = {"L": L, "Rmax": R, "N": N, "h": h}
    pots = {"coulomb10": np.concatenate([[0.0], -10.0 / rr[1:]]), "screened18": screened_potential(rr, 18.0)}
    for pname, V in pots.items():
        hd = r.ref_unumerov_create(O.dp(V), N, R)
        Zp = 10.0 if pname.endswith("10") else 18.0
        Es = np.concatenate([-rng.uniform(1e-3, Zp * Zp + 1, 16), -10.0 ** rng.uniform(-3, 1, 6), [-(Zp ** 2) / 2, -(Zp ** 2) / 8, -33.0, -31.0, 0.5, 50.0]])
        rows, ms = [], []
        P = np.zeros(N)
        for l in range(4):
            for E in Es:
                u0 = r.ref_usolution_in_zero(hd, l, float(E))
                for lim in (0, 2, 5):
                    rows.append([l, E, lim, r.ref_ucount_nodes(hd, l, float(E), lim), u0])
                mp_ = r.ref_umatch(hd, l, float(E), O.dp(P))
                ms.append(np.concatenate([[l, E, mp_, np.nansum(P), np.nansum(np.abs(P))], P[:: N // 64][:64]]))
        out["sweeps_" + pname] = np.array(rows)
        out["match_" + pname] = np.array(ms)
        r.ref_unumerov_destroy(hd)
    # level driver on the screened potential (Ar configuration)
    V = pots["screened18"]
    lv = O.subshells(18)
    hd = r.ref_unumerov_create(O.dp(V), N, R)
    n = np.array([a for a, _, _ in lv], np.int32)
    l = np.array([b for _, b, _ in lv], np.int32)
    occ = np.array([c for _, _, c in lv], np.int32)
    E = np.zeros(len(lv))
    nd = np.zeros(N)
    Eel, Bot = C.c_double(0), C.c_double(-18.0 * 18 - 1.0)
    conv = r.ref_uloop_over_levels(hd, len(lv), O.ip(n), O.ip(l), O.ip(occ), O.dp(E), O.dp(nd), C.byref(Eel), C.byref(Bot))
    out["levels_E"] = E
    out["levels_newdensity_sample"] = nd[:: N // 256].copy()
    out["levels_scalars"] = np.array([Eel.value, Bot.value, conv, nd.sum()])
    r.ref_unumerov_destroy(hd)
    # Poisson
    Lp, Rp_ = 12, 25.0
    Np = r.ref_num_nodes(Lp)
    rp = (Rp_ / (Np - 1)) * np.arange(Np)
    for tag, Z in (("Z2", 2), ("Z18", 18)):
        rho = Z * np.exp(-2 * rp) / np.pi
        q = r.ref_poisson_create(Lp, 0.0)
        U = np.zeros(Np)
        r.ref_solve_poisson_uniform(q, Z, Rp_, O.dp(rho), Np, O.dp(U))
        out["poisson_%s_U" % tag] = U
        r.ref_poisson_destroy(q)
    meta["poisson_grid"] = {"L": Lp, "Rmax": Rp_}
    # Chachiyo functional (ExcCor.h), both parameter sets, on a density ladder
    nl = np.concatenate([10.0 ** np.linspace(-20, 6, 131), [0.0, 9.99e-19, 1e-18]])
    out["chachiyo_n"] = nl
    for imp in (0, 1):
        v, e = np.zeros_like(nl), np.zeros_like(nl)
        r.ref_chachiyo(imp, O.dp(nl), O.dp(v), O.dp(e), len(nl))
        out["chachiyo_%d" % imp] = np.stack([v, e])
    np.savez_compressed(os.path.join(HERE, "uniform.npz"), **out)
    # end to end: Ne LDA and N LSDA on 4097 uniform nodes, Rmax 15
    e2e = {}
    for mode, Z, tag in ((2, 10, "Ne_uLDA_L12"), (3, 7, "N_uLSDA_L12")):
        txt = run_ref(mode, Z, 12, 0.5, 15.0, 0.0)
        steps = parse_run(txt)
        e2e[tag] = {"Z": Z, "L": 12, "Rmax": 15.0, "nsteps": len(steps), "finished": "Finished!" in txt, "steps": steps[:3] + [steps[-1]],
                    "etotal_all": [s["energies"][0] for s in steps], "banner": txt.splitlines()[0],
                    "tail": txt.strip().splitlines()[-2:]}
        print(tag, len(steps), steps[-1]["energies"], flush=True)
    meta["end_to_end"] = e2e
    with open(os.path.join(HERE, "uniform_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "table"
    jobs = int(sys.argv[sys.argv.index("--jobs") + 1]) if "--jobs" in sys.argv else 6
    if what == "table":
        table(jobs)
    elif what == "uniform":
        uniform()
    elif what == "l20cond":
        l20cond()
    elif what == "jitter":
        jitter(min(jobs, 4))
    else:
        l20()
