#!/usr/bin/env python3
"""bench.py -- SCF-step wall time, Numerov sweeps/s and Poisson V-cycles/s for Radon (Z=86) at 131073 grid points on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--atoms B] [--lsda] [--levels 17] [--no-cpu] [--no-extras]

Workload of the headline line (BASELINE.json configs[1]): Rn Z=86 LDA, 17 multigrid levels (131073 nodes), delta = 1e-4,
Rmax = 50, mixing 0.5 (README.md:54 of the reference); one "step" = one SCF iteration of the atom batch, state resident in
HBM: level search for all 15 (n,l) subshells (speculative bisection trees of Numerov sweeps, un-chained brackets), density
mixing, multigrid Poisson (100 V-cycles), VWN and the five Simpson-3/8 energy integrals.  The timed steps continue the SCF
iteration from the warm-up steps (synthetic start: the reference's flat density).

value = reference-equivalent Numerov sweeps per second over the WHOLE step wall time: the sweeps the reference's own
bisection path needs for these steps (CountNodes + SolutionInZero + Match; `sweeps_reference` counts every call the reference
makes, including the ~52 CountNodes calls per node-less level whose outcome -- "count < 0" -- is decided here without
integrating; `sweeps_reference_executed` leaves those out) divided by the elapsed time including Poisson / XC / integrals:
the same quantity the CPU baseline reports.

The ONE line on stdout is compact (< 7 KB: the driver keeps the last 8 KB of stdout; compact_line()); the complete result with every
kernel object and its notes goes to profiles/bench_full_last.json (and gpurun_out/bench_full_last.json).  On the line:
  roofline       the DOMINANT kernel of the timed region (by HIP-event time): the persistent multigrid kernel or the sweep
                 kernel; achieved = algorithmic bytes / its HIP-event time, against 8 TB/s; traffic,
                 hbm_GBps_counters, frac_counters = rocprofv3 FETCH/WRITE bytes of the committed profile of this workload
  kernels        both hot kernels, flat: frac (reference's bisection path), frac_issued (every speculative trial), counters
  extra          (N = 1 only) one flat object per further workload: tolerance mode, Rn LSDA (configs[2]), a 256-atom batch,
                 1 048 577 nodes x 1 and x 16 atoms (configs[4]); --all-extras: LSDA tolerance, 1024 atoms, dense-K sweeps
  cpu_baseline   the oracle on the host: 1 core (how the reference runs) + scalars for the table variant, 15 level threads and
                 one replica per physical core

N > 1 (launched by torch.distributed.run): atoms are independent, so every rank advances its own replica of the batch (weak
scaling, no data-path collective); the per-atom result records are all_gathered over RCCL once after the last step, inside
the timed region.
"""
import argparse
import ctypes as C
import glob
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec); the attainable figure is measured
HBM_MEASURED = {"copy": None, "triad": None}   # live by a copy / triad kernel of the library (dfta_ctx_measure_hbm) at the start of the run
NUMEROV_BYTES_PER_POINT = 8    # one fp64 V_i per traversed grid point per trial
POISSON_BYTES_PER_VCYCLE = {14: 6162448, 17: 49285736, 20: 394267840}   # SURVEY.md section 8(d), every pass counted
# fp64 VALU issue: one wave64 instruction per SIMD every 1.86 ns (profiles/microbench/issue_rate.hip), 256 CUs x 4 SIMDs
VALU_WAVE_INSTR_PER_S = 256 * 4 / 1.86e-9
# wave-level VALU instructions per grid point of one 64-trial block in k_sweep_pipe (numerov.hip): producers 15 (f: 4, d: 2,
# reciprocal: 5, lane broadcasts: 4), integrator 8 (+1 for the sign word of a COUNT sweep), counter < 1
SWEEP_VALU_PER_BLOCK_POINT = 24
GRIDS = {14: (5e-4, 25.0), 17: (1e-4, 50.0), 20: (1.25e-5, 50.0)}


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline: the oracle (test infrastructure, used here as the measured CPU leg only)
# ---------------------------------------------------------------------------------------------------------------
def cpu_worker(levels, lsda, steps, tables, threads=1):
    """runs in its own process: `steps` SCF steps of Rn on the oracle, prints one JSON line.  threads > 1: the level-parallel
    variant (un-chained clamped brackets -- the GPU path's mode -- one OpenMP thread per level, oracle/libdfta_oracle_omp.so)"""
    if threads > 1:
        os.environ["DFTA_ORACLE_OMP"] = "1"
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    delta, rmax = GRIDS.get(levels, (1e-4, 50.0))
    o = O.oracle()
    if threads > 1:
        o.dfo_set_level_threads(threads)
    t0 = time.time()
    s = o.dfo_scf_create(int(lsda), 86, levels, 0.5, rmax, delta, 3 if threads > 1 else 1)
    t_setup = time.time() - t0
    if tables:
        o.dfo_tables_enable(C.byref(s.contents.g))
    e = O.Energies()
    sweeps = 0
    vcycles0 = s.contents.ps.contents.n_vcycles
    t0 = time.time()
    for _ in range(steps):
        o.dfo_scf_step(s, C.byref(e))
        for arr, n in ((s.contents.la, s.contents.nla), (s.contents.lb, s.contents.nlb if lsda else 0)):
            for i in range(n):
                sweeps += arr[i].n_count + arr[i].n_zero + 1
    dt = time.time() - t0
    vc = s.contents.ps.contents.n_vcycles - vcycles0
    o.dfo_tables_disable()
    o.dfo_scf_destroy(s)
    print(json.dumps({"sweeps": sweeps, "seconds": dt, "vcycles": vc, "setup_s": t_setup, "etotal": e.Etotal}))


def _start_cpu(levels, lsda, steps, tables, n, threads=1):
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(levels), str(int(lsda)), str(steps), str(int(tables)), str(threads)]
    return [subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True) for _ in range(n)]


def _join_cpu(procs):
    return [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in procs]


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """distinct (package, core) pairs of /proc/cpuinfo; falls back to the logical CPU count"""
    try:
        cores, phys = set(), None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                cores.add((phys, ln.split(":", 1)[1].strip()))
        if cores:
            return len(cores)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_baseline_start(levels, lsda, steps):
    """The oracle (plain-C restatement of the reference) timed on this host, in child processes.  Three variants start at once (1 + 1 + 15
    threads on a host with far more cores; the GPU extras run meanwhile and need one host thread): 1 core as the reference runs, 1 core with
    tables, one atom with its levels on 15 OpenMP threads (SURVEY 8d ii).  The all-core replicas run alone afterwards (cpu_baseline_finish)."""
    nlev = 15                                                              # Rn: 15 subshells per spin
    return {"levels": levels, "lsda": lsda, "steps": steps, "nlev": nlev,
            "one": _start_cpu(levels, lsda, steps, False, 1), "tab": _start_cpu(levels, lsda, steps, True, 1),
            "par": _start_cpu(levels, lsda, steps, False, 1, threads=nlev)}


def cpu_baseline_finish(job, all_cores=True):
    levels, lsda, steps, nlev = job["levels"], job["lsda"], job["steps"], job["nlev"]
    nproc = os.cpu_count() or 1
    one, tab, par = _join_cpu(job["one"]), _join_cpu(job["tab"]), _join_cpu(job["par"])
    o, t = one[0], tab[0]
    assert abs(o["etotal"] - t["etotal"]) == 0.0                     # the table variant is bit-identical
    tag = "Rn %s @ %d levels" % ("LSDA" if lsda else "LDA", levels)
    out = {"value": o["sweeps"] / o["seconds"], "unit": "sweeps/s", "cores": 1, "kind": "port",
           "sample": "%d SCF steps of %s on the oracle (oracle/dfta_oracle.c, gcc -O2 -ffp-contract=off, 1 thread): %.2f s, %d sweeps, "
                     "%d V-cycles; setup (flat density + Poisson) %.2f s excluded" % (steps, tag, o["seconds"], o["sweeps"], o["vcycles"], o["setup_s"]),
           "ms_per_step": 1e3 * o["seconds"] / steps, "vcycles_per_s": o["vcycles"] / o["seconds"],
           "cpu_model": cpu_model(), "nproc": nproc,
           # not timed alone: two more CPU variants (1 + %d threads) and the GPU extras' host thread run on the same host meanwhile
           "concurrent_with": "the table variant (1 thread), the level-parallel variant (%d threads) and the host thread of the GPU extras, on %d logical CPUs"
                              % (nlev, nproc),
           "table_variant": {"value": t["sweeps"] / t["seconds"], "unit": "sweeps/s", "cores": 1, "ms_per_step": 1e3 * t["seconds"] / steps,
                             "note": "r_i and exp(2 i delta) looked up instead of re-evaluated per point; results bit-identical"},
           "level_parallel": {"value": par[0]["sweeps"] / par[0]["seconds"], "unit": "sweeps/s", "cores": nlev,
                              "ms_per_step": 1e3 * par[0]["seconds"] / steps,
                              "note": "ONE atom, the levels of a spin on %d OpenMP threads (un-chained clamped brackets: the GPU path's mode; "
                                      "bit-identical to its serial form), multigrid / XC / integrals serial: the per-atom latency a CPU can reach "
                                      "(SURVEY 8d ii)" % nlev}}
    if all_cores:
        nall = max(1, min(physical_cores(), len(os.sched_getaffinity(0))))     # every physical core this process may use
        steps_all = 1                                                           # one step per replica: 128 concurrent replicas share the memory system
        allc = _join_cpu(_start_cpu(levels, lsda, steps_all, False, nall))
        out["all_cores"] = {"value": sum(x["sweeps"] for x in allc) / max(x["seconds"] for x in allc), "unit": "sweeps/s", "cores": nall,
                            "vcycles_per_s": sum(x["vcycles"] for x in allc) / max(x["seconds"] for x in allc),
                            "physical_cores": physical_cores(),
                            "note": "%d independent replicas of the same run, one process per physical core (atoms are the parallel axis of the "
                                    "reference's algorithm; the reference itself is single-threaded), %d step(s) each, slowest replica's time. "
                                    "Inside one atom only the level loop parallelises (15 subshells, ~77 %% of a step; the multigrid is a serial "
                                    "recurrence): at most ~3.5x per atom by Amdahl, so replicas are the all-core figure that favours the CPU"
                                    % (nall, steps_all)}
    return out


def cpu_baseline(levels, lsda, steps, all_cores=True):
    return cpu_baseline_finish(cpu_baseline_start(levels, lsda, steps), all_cores)


# ---------------------------------------------------------------------------------------------------------------
# profiles
# ---------------------------------------------------------------------------------------------------------------
def source_sha():
    h = hashlib.sha256()
    src = os.path.join(ROOT, "dftatom_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h")) + glob.glob(os.path.join(src, "*.inc"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, workload="default"):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC passes of the same workload
    (profiles/*<workload>_hbm_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate passes; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  bench.py cannot run the profiler on itself: the number is the measured one of
    that profile, returned with its tag -- and withheld (null) when the kernels' sources have changed since it was taken."""
    import re
    pat = re.compile(r"^r\d+[a-z]?_%s_hbm_traffic\.json$" % re.escape(workload))       # r05_scan_tol_..., not r05_batch256_scan_tol_...
    files = sorted((f for f in glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")) if pat.match(os.path.basename(f))),
                   key=lambda f: (os.path.basename(f).split("_")[0], os.path.getmtime(f)))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        tag = {"profile": os.path.basename(files[-1]), "source_sha": d.get("source_sha"), "current": d.get("source_sha") == source_sha()}
        if not tag["current"]:
            return None, tag
        names = [kernel] + (["k_scan_levels_group"] if kernel == "k_scan_levels" else [])     # one atom: a group of workgroups per level
        for nm in reversed(names):
            if nm in d["kernels"]:
                return float(d["kernels"][nm]["hbm_bytes_per_launch_fetch_doubled"]), tag
        return None, tag
    except Exception:
        return None, None


def counter_rate(d, kernel, workload):
    """adds the counter-based HBM rate of a kernel object: (2 x FETCH_SIZE + WRITE_SIZE) per launch of the committed profile of
    this workload / the launch time measured in this run.  This -- not the algorithmic figure -- is a bandwidth."""
    traffic, tag = pmc_traffic(kernel, workload)
    d["hbm_bytes_per_launch_counters"] = traffic
    d["counters_profile"] = tag
    ms = d.get("avg_launch_ms")
    d["hbm_GBps_counters"] = traffic / (ms * 1e-3) / 1e9 if traffic and ms else None
    d["frac_counters"] = d["hbm_GBps_counters"] / HBM_PEAK_GBS if d["hbm_GBps_counters"] else None


def rocprof_launch(kernel, workload="default"):
    """average launch duration of `kernel` in the newest committed rocprofv3 kernel trace of the same workload
    (profiles/<round>_<workload>_bench_kernel_stats.csv, profiles/collect.sh + summarize.py), over the TIMED launches of that run where the
    summary has them (a kernel launched once per SCF step), with the window it describes -- so that the line's HIP-event average can be
    checked against the profiler's (tests/test_bench_line.py).  (None, None) without a profile."""
    import csv
    import re
    pat = re.compile(r"^r\d+[a-z]?_%s_bench_kernel_stats\.csv$" % re.escape(workload))
    files = sorted((f for f in glob.glob(os.path.join(ROOT, "profiles", "*_bench_kernel_stats.csv")) if pat.match(os.path.basename(f))),
                   key=lambda f: (os.path.basename(f).split("_")[0], os.path.getmtime(f)))
    if not files:
        return None, None
    try:
        with open(files[-1]) as fh:
            rows = {r["kernel"]: r for r in csv.DictReader(fh)}
        r = rows.get(kernel)
        if r is None:
            return None, None
        side = files[-1].replace("_bench_kernel_stats.csv", "_bench_under_rocprof.json")
        wu = st = None
        if os.path.exists(side):
            with open(side) as fh:
                b = json.load(fh)
            wu, st = b.get("warmup"), b.get("steps")
        if r.get("timed_avg_us"):
            return float(r["timed_avg_us"]) / 1e3, {"profile": os.path.basename(files[-1]), "launches": int(r["timed_calls"]),
                                                    "window": "the timed launches only: SCF steps %s..%s of `bench.py --steps %s --warmup %s` under rocprofv3 --kernel-trace"
                                                              % (wu, (wu + st - 1) if wu is not None and st else None, st, wu)}
        return float(r["avg_us"]) / 1e3, {"profile": os.path.basename(files[-1]), "launches": int(r["calls"]),
                                          "window": "ALL launches of `bench.py --steps %s --warmup %s` under rocprofv3 --kernel-trace (warm-up included)" % (st, wu)}
    except Exception:
        return None, None


def workload_tag(levels, atoms, lsda, poisson_mode, sweep_mode):
    """name of the committed rocprofv3 profile (profiles/<round>_<tag>_hbm_traffic.json, profiles/collect.sh) of a workload, or None"""
    mode = {("exact", "exact"): "", ("tolerance", "exact"): "tolerance", ("exact", "tolerance"): "scan", ("tolerance", "tolerance"): "scan_tol",
            ("adaptive", "tolerance"): "scan_adaptive"}.get((poisson_mode, sweep_mode))
    if mode is None:
        return None
    if levels == 17 and atoms == 1:
        base = "rn_lsda" if lsda else "default"
        if lsda:
            return base if mode == "" else None
        return mode or base
    if levels == 17 and atoms == 256 and not lsda:
        return "batch256" + ("_" + mode if mode else "")
    if levels == 17 and atoms == 12 and not lsda and mode == "":
        return "batch12"                 # a shard-sized batch: the multigrid's 17-workgroup resident groups
    if levels == 20 and lsda and atoms in (1, 16):
        base = "l20" if atoms == 1 else "l20_batch16"
        return base + ("_" + mode if mode else "")
    return None


# ---------------------------------------------------------------------------------------------------------------
# what the tests assert for the mode a number was measured in (so that a reader of the line sees what kind of number it is)
# ---------------------------------------------------------------------------------------------------------------
def parity_gates(poisson_mode, sweep_mode):
    """poisson_mode / sweep_mode: 'exact', 'tolerance', 'adaptive'.  Values = the gates of tests/ against the compiled reference's goldens
    (test_gpu_parity / test_gpu_configs: exact; test_gpu_scan, test_gpu_resident, test_frontend::test_readme_tables_in_the_opt_in_modes: opt-in)."""
    exact_sweeps = sweep_mode == "exact"
    g = {"node_counts": "bit-exact (counts, cut-offs, loop trips = the compiled reference's)" if exact_sweeps else
                        "exact outside the round-off band: decisions identical while the interval is wider than 1.5e-10|E|+2e-9 Ha",
         "per_level_dE": "2e-12 Ha (the reference's midpoints, same sweep counts)" if exact_sweeps else "6e-11|E|+6e-10 Ha vs the exact path (same V)",
         "converged_dE": "README six decimals (Ar 5, Rn 15 eigenvalues); 2e-7 Ha+1e-10|E| vs the reference's last steps",
         "etotal_rel": 1e-9 if (exact_sweeps and poisson_mode == "exact") else 2e-9,
         "vcycles_per_solve": 100 if poisson_mode != "adaptive" else "6-8 (stops on the round-off floor; the reference runs 100)"}
    if poisson_mode != "exact":
        g["poisson_U"] = "2e-9 Z vs the exact solve"
    return g


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 4: the periodic-table sweep, sharded by atom (dftatom_amd/sweep.py), one all_gather of the records
# ---------------------------------------------------------------------------------------------------------------
def per_call_surface():
    """The reference's orchestration asks for ONE trial energy per call (SolveSchrodingerCountNodes / SolutionInZero).  dftatom_amd/compat/percall_levels
    restates that call stream (LocateInterval + the u(0) bisection, DFTAtom.cpp:493-604) on DFT::Numerov's per-call surface for the 15 levels of Rn
    on a bare Coulomb potential -- exact kernels; compat/call_stream.h serves most calls from what it had integrated ahead.  A child process (its own
    HIP context; started after the timed region).  None when the binary is not there."""
    import re
    import subprocess
    exe = os.path.join(ROOT, "dftatom_amd", "compat", "percall_levels")
    if not os.path.exists(exe):
        return None

    def run(args, env):
        out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, env=dict(os.environ, **env)).stdout
        m = re.search(r"calls (\d+) launches (\d+) hits (\d+) seconds (\S+)", out)
        return (int(m.group(1)), int(m.group(2)), float(m.group(4))) if m else None
    try:
        big = run(["86", "17", "0.0001", "50", "15"], {})
        small = run(["86", "14", "0.0005", "25", "15"], {})
        plain = run(["86", "14", "0.0005", "25", "15"], {"DFTA_COMPAT_NOSPECULATE": "1"})
    except Exception as e:                      # must not cost the line
        return {"error": repr(e)[:120]}
    if not (big and small and plain):
        return None
    return {"what": "Rn's 15 levels searched through DFT::Numerov's per-call surface (the reference's call stream), exact kernels, Coulomb potential",
            "calls": big[0], "launches_131073": big[1], "seconds_131073": big[2], "ms_per_call_131073": 1e3 * big[2] / big[0],
            "launches_16385": small[1], "seconds_16385": small[2], "seconds_16385_one_trial_per_call": plain[2],
            "speedup_16385": plain[2] / small[2] if small[2] > 0 else None}


def periodic_table(D, ctx, grid, levels, world, rank, dist, torch, shared, zmax, max_steps=100):
    """every rank advances its partition_atoms shard of Z = 1..zmax to the reference's stop test (or its 100-step cap), then ONE all_gather
    of the fixed-size records (RCCL; gloo in the shared-GPU test mode).  Returns (on every rank) the whole-job wall time = max over ranks."""
    from dftatom_amd import sweep
    Zs = list(range(1, zmax + 1))
    shards = sweep.partition_atoms(Zs, world)
    mine = shards[rank]
    cap = max(len(sh) for sh in shards)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.time()
    steps = 0
    if mine:
        scf = D.Scf(ctx, grid, mine, lsda=False)
        while steps < max_steps:
            scf.step(want_stats=False)
            steps += 1
            _, fin = scf.energies()
            if fin.all():
                break
    block = torch.zeros((cap, D.RECORD_DOUBLES), dtype=torch.float64, device="cuda")
    if mine:
        scf.records_into(block.data_ptr())
    ctx.synchronize()
    t_mine = time.time() - t0
    table = sweep.gather_records(block.cpu() if shared else block, dist if world > 1 else None)
    torch.cuda.synchronize()
    t_all = time.time() - t0
    if mine:
        scf.close()
    times = [t_mine]
    if world > 1:
        tt = torch.tensor([t_mine, t_all], dtype=torch.float64, device="cpu" if shared else "cuda")
        allt = [torch.empty_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        times = [float(x[0]) for x in allt]
        t_all = max(float(x[1]) for x in allt)
    rows = [sweep.record_fields(table[z]) for z in sorted(table)]
    return {"workload": "Z = 1..%d LDA @ %d levels (%d pts), atoms sharded over %d rank(s) by predicted shard time, every atom to the "
                        "reference's stop test or its 100-step cap, one all_gather of %d doubles per atom" % (zmax, levels, grid.N, world, D.RECORD_DOUBLES),
            "seconds": t_all, "shard_seconds": times, "slowest_rank": int(max(range(len(times)), key=lambda k: times[k])),
            "atoms_per_rank": [len(sh) for sh in shards], "atoms": len(rows), "finished": int(sum(r["finished"] for r in rows)),
            "atom_steps": int(sum(r["steps"] for r in rows)), "mode": "exact kernels (default path)",
            "etotal_rn": next((r["Etotal"] for r in rows if r["Z"] == 86), None),
            "measured": "this run, %d GPU(s)" % world}


# ---------------------------------------------------------------------------------------------------------------
# one measured workload
# ---------------------------------------------------------------------------------------------------------------
def run_workload(D, ctx, grid, levels, atoms, lsda, steps, warmup, tree_depth, barrier, torch, after_steps=None, poisson_mode=None, sweep_mode=None):
    scf = D.Scf(ctx, grid, [86] * atoms, lsda=lsda, alpha=0.5, levels_mode=D.LEVELS_BATCHED, tree_depth=tree_depth,
                poisson_mode=D.POISSON_EXACT if poisson_mode is None else poisson_mode,
                sweep_mode=D.SWEEPS_EXACT if sweep_mode is None else sweep_mode)
    for _ in range(warmup):
        scf.step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    keys = ("sweeps_issued", "sweeps_reference", "sweeps_reference_executed", "points_traversed", "points_reference", "vcycles", "rounds",
            "ms_sweep_kernels", "ms_levels", "ms_poisson", "ms_tail")
    tot = {k: 0 for k in keys}
    t0 = time.time()
    ev0.record()
    for _ in range(steps):
        st = scf.step()
        for k in keys:
            tot[k] += getattr(st, k)
    if after_steps:
        after_steps(scf)
    ev1.record()
    barrier()
    tot["elapsed"] = time.time() - t0
    tot["ev_ms"] = ev0.elapsed_time(ev1)
    tot["steps"] = steps
    tot["trials_per_round"] = scf.trials_per_round
    tot["tree_depth"] = scf.tree_depth
    tot["levels_layout"] = {0: "one block of 2^depth trials per job", 1: "latency mode (slots re-allotted every round)",
                            2: "packed rounds (depth chosen per round, floor = tree depth)", 3: "latency mode over the live jobs",
                            4: "scan sweeps (tolerance mode: one workgroup per level, no rounds)",
                            5: "device-side exact search (one persistent kernel, every level at its own pace)",
                            6: "own-pace batch search (one workgroup per level in one launch; opt-in)"}.get(int(st.levels_layout), "?")
    tot["scan"] = int(st.levels_layout) == 4
    tot["persist"] = int(st.levels_layout) == 5
    tot["poisson_G"] = scf.poisson_info()[0]
    tot["energies"] = scf.energies()[0][0].as_list()
    return scf, tot


def kernel_figures(tot, levels, N, atoms, workload=None):
    """per-kernel roofline figures of a workload (HIP-event times measured inside the library on the launch stream).
    `algorithmic_*` = SURVEY 8d bytes / time: a yardstick, NOT a bandwidth (levels resident in LDS / L2 and trials sharing a table row
    make it exceed what HBM moves); `hbm_GBps_counters` = rocprofv3 FETCH/WRITE bytes of the committed profile of `workload` / time."""
    forced = next((e.split("=", 1)[1] for e in os.environ.get("DFTA_DEBUG", "").split(",") if e.startswith("SWEEP_KERNEL=")), "")
    # which sweep kernel the rounds ran: the library picks per launch by the blocks of the round (<= 768: pipelined); packed rounds lay out
    # fewer trials than the solver has room for, so the average issued trials per round decide here, not the capacity
    per_round = tot["sweeps_issued"] / max(tot["rounds"], 1) if tot.get("rounds") else tot["trials_per_round"]
    piped = forced == "pipe" or (forced != "fused" and min(tot["trials_per_round"], per_round) // 64 <= 768)
    queued = "LEVELS_NOQUEUE" not in os.environ.get("DFTA_DEBUG", "")      # the fused sweeps of a batch are launched longest block first (k_sweep_queue)
    sname = "k_scan_levels" if tot.get("scan") else ("k_levels_persist" if tot.get("persist") else ("k_sweep_pipe" if piped else ("k_sweep_queue" if queued else "k_sweep")))
    t_sw = tot["ms_sweep_kernels"] * 1e-3
    # host rounds: one launch per round; the device-side search: ONE launch per SCF step (sweeps of every round, walk, match, normalisation)
    launches = max(tot["steps"], 1) if tot.get("persist") else max(tot["rounds"], 1)
    b_issued = NUMEROV_BYTES_PER_POINT * tot["points_traversed"]
    b_ref = NUMEROV_BYTES_PER_POINT * tot["points_reference"]
    block_points = tot["points_traversed"] / 64.0       # lower bound: 64 live lanes in every block
    sweep = {"kernel": sname, "what": "Numerov CountNodes / SolutionInZero sweeps", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "launches": launches, "avg_launch_ms": tot["ms_sweep_kernels"] / launches,
             "algorithmic_GBps_issued": b_issued / t_sw / 1e9 if t_sw else None, "frac_issued": b_issued / t_sw / 1e9 / HBM_PEAK_GBS if t_sw else None,
             "algorithmic_GBps": b_ref / t_sw / 1e9 if t_sw else None,
             "frac": b_ref / t_sw / 1e9 / HBM_PEAK_GBS if t_sw else None,
             "bytes_per_launch_issued": b_issued / launches, "bytes_per_launch": b_ref / launches,
             "binding_resource": ("fp64 VALU issue of one compute unit per level (transfer-matrix scan of one trial by 512 lanes, ~150 sweeps back to back)"
                                  if tot.get("scan") else
                                  "fp64 VALU issue of the integrator wave (one block of 64 trials per CU; sequential three-term recurrence), "
                                  "x the ~7 dependent rounds of a level's three bisections inside the one launch" if tot.get("persist") else
                                  "fp64 VALU issue of the integrator wave (one block of 64 trials per CU; sequential three-term recurrence)" if piped else
                                  "fp64 VALU issue of the SIMDs (one wave per block of 64 trials, two waves per SIMD, 24 instructions per point; a wave "
                                  "runs as long as its longest lane -- the lane-based valu_issue figure undercounts the busy time)"),
             "valu_issue": {"wave_instr_per_block_point": SWEEP_VALU_PER_BLOCK_POINT, "block_points_per_s": block_points / t_sw if t_sw else None,
                            "ceiling_wave_instr_per_s": VALU_WAVE_INSTR_PER_S,
                            "frac": SWEEP_VALU_PER_BLOCK_POINT * block_points / t_sw / VALU_WAVE_INSTR_PER_S if t_sw else None,
                            "ns_per_point_per_block": 1e9 * t_sw / (max(tot["rounds"], 1) * N) if t_sw else None,
                            "rounds_per_step": tot["rounds"] / max(tot["steps"], 1),
                            "note": "static instruction count from numerov.hip x measured points; profiles/*_sq_counters.json holds SQ_INSTS_VALU"},
             "note": "8 B per traversed grid point per trial (SURVEY 8d). frac = only the trials on the reference's bisection path (what its "
                     "sequential loop integrates); frac_issued = every trial of the speculative trees. The 64 trials of a block share "
                     "one table row, so physical HBM traffic (hbm_GBps_counters) is far below either figure."}
    t_ps = tot["ms_poisson"] * 1e-3
    b_ps = POISSON_BYTES_PER_VCYCLE.get(levels, 376 * N) * tot["vcycles"]
    psolves = max(tot["steps"], 1)
    pname = "k_poisson_solve_res" if tot["poisson_G"] >= 33 else "k_poisson_solve"
    pois = {"kernel": pname, "what": "persistent multigrid: FMG ramp + 100 V-cycles per launch",
            "bound": "hbm", "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "launches": psolves, "avg_launch_ms": tot["ms_poisson"] / psolves,
            "algorithmic_GBps": b_ps / t_ps / 1e9 if t_ps else None, "frac": b_ps / t_ps / 1e9 / HBM_PEAK_GBS if t_ps else None,
            "bytes_per_launch": b_ps / psolves, "vcycles_per_s": tot["vcycles"] / t_ps if t_ps else None,
            # "a fully fused cycle's compulsory traffic is only 24 N": read S, read and write Phi of the finest level once per V-cycle
            "frac_compulsory": 24.0 * N * tot["vcycles"] / t_ps / 1e9 / HBM_PEAK_GBS if t_ps else None,
            "workgroups": atoms * tot["poisson_G"],
            "binding_resource": ("latency of the ordered Gauss-Seidel recurrence: %d atom(s) x %d workgroups on 256 CUs; " % (atoms, tot["poisson_G"])) +
                                ("resident groups keep the finest levels in LDS (one fused 3-sweep pass of 112+C+2 dependent "
                                 "steps and one exchange per visit), a coarse workgroup runs the levels below" if tot["poisson_G"] >= 33 else
                                 "every sweep %d+95 dependent steps" % max(1, (N - 1) // (256 * tot["poisson_G"]))),
            "note": "algorithmic bytes per V-cycle with every pass counted (GS 24 B/pt x 3 sweeps per visit, restrict, prolong: SURVEY 8d); the "
                    "level storage of one atom (6.3 MB at 17 levels) stays in LDS / L2 and fused visits read a level once per three sweeps, so "
                    "the figure can exceed the HBM peak: hbm_GBps_counters is the bandwidth"}
    for d, k in ((sweep, sname), (pois, pname)):
        d["peak_measured_copy"] = HBM_MEASURED["copy"]
        if workload:
            counter_rate(d, k, workload)
            d["avg_launch_ms_rocprof"], d["rocprof_window"] = rocprof_launch(k, workload)
    return sweep, pois


def summarize(tot, levels, N, atoms, lsda, world, delta, rmax, workload=None):
    sweep, pois = kernel_figures(tot, levels, N, atoms, workload)
    return {"workload": "Rn Z=86 %s, %d multigrid levels (%d pts), delta=%g, Rmax=%g, mixing 0.5, %d atom(s)/GPU, un-chained clamped brackets, "
                        "tree depth %d, %s" % ("LSDA" if lsda else "LDA", levels, N, delta, rmax, atoms, tot["tree_depth"], tot["levels_layout"]),
            "sweeps_executed_per_s": tot["sweeps_reference_executed"] / tot["elapsed"],
            "sweeps_reference_equivalent_per_s": tot["sweeps_reference"] / tot["elapsed"],
            "issued_per_useful": tot["sweeps_issued"] / max(tot["sweeps_reference_executed"], 1),
            "sweeps_issued_per_s": tot["sweeps_issued"] / tot["elapsed"], "vcycles_per_s": tot["vcycles"] / tot["elapsed"],
            "ms_per_step": 1e3 * tot["elapsed"] / tot["steps"], "ms_per_atom_step": 1e3 * tot["elapsed"] / tot["steps"] / atoms,
            "steps": tot["steps"], "rounds_per_step": tot["rounds"] / tot["steps"], "poisson_workgroups_per_atom": tot["poisson_G"],
            "phase_ms_per_step": {"levels": tot["ms_levels"] / tot["steps"], "poisson": tot["ms_poisson"] / tot["steps"],
                                  "tail": tot["ms_tail"] / tot["steps"], "hip_event_total": tot["ev_ms"] / tot["steps"]},
            "kernels": {"sweep": sweep, "poisson": pois}}


# ---------------------------------------------------------------------------------------------------------------
# the ONE line on stdout: compact (the driver keeps the last 8 KB of stdout); everything else goes to a side file
# ---------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 7000


def _r(x, sig=6):
    """numbers to `sig` significant digits (the full values are in the side file)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        return float("%.*g" % (sig, x))
    return x


def _flat_kernel(k, keys):
    return {a: _r(k.get(a)) for a in keys if a in k}


def compact_line(full):
    """the driver's schema keys + roofline (dominant kernel) + cpu_baseline + one flat object per extra workload.  Pure function of the
    full result (tests/test_bench_line.py feeds it a canned one)."""
    sweep_keys = ("kernel", "launches", "avg_launch_ms", "frac", "frac_issued", "bytes_per_launch", "hbm_GBps_counters", "frac_counters")
    pois_keys = ("kernel", "launches", "avg_launch_ms", "frac", "frac_compulsory", "bytes_per_launch", "vcycles_per_s", "workgroups", "hbm_GBps_counters", "frac_counters")
    out = {k: _r(full[k]) if not isinstance(full[k], (dict, list)) else full[k]
           for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                     "data", "config") if k in full}
    for k in ("ranks_seen", "value_reference_equivalent", "sweeps_issued_per_s", "poisson_vcycles_per_s", "poisson_vcycles_per_s_kernel", "rounds_per_step",
              "device", "compute_units"):
        if k in full:
            out[k] = _r(full[k])
    out["phase_ms_per_step"] = {k: _r(v, 5) for k, v in full.get("phase_ms_per_step", {}).items()}
    out["etotal_last_step"] = (full.get("energies_last_step") or [None])[0]
    rf = full["roofline"]
    out["roofline"] = {k: _r(rf.get(k)) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_issued", "traffic", "bytes_per_launch",
                                                   "avg_launch_ms", "launches", "hbm_GBps_counters", "frac_counters", "peak_measured_copy",
                                                   "achieved_is", "avg_launch_ms_rocprof") if k in rf}
    if isinstance(rf.get("rocprof_window"), dict):
        out["roofline"]["rocprof_window"] = "%s: %s (%s launches)" % (rf["rocprof_window"].get("profile"), rf["rocprof_window"].get("window"), rf["rocprof_window"].get("launches"))
    ks = full.get("kernels", {})
    out["kernels"] = {"sweep": _flat_kernel(ks.get("sweep", {}), sweep_keys), "poisson": _flat_kernel(ks.get("poisson", {}), pois_keys)}
    if "cpu_baseline" in full:
        c = full["cpu_baseline"]
        out["cpu_baseline"] = {k: _r(c.get(k)) for k in ("value", "unit", "cores", "kind", "ms_per_step", "vcycles_per_s", "cpu_model", "nproc", "concurrent_with") if k in c}
        out["cpu_baseline"]["sample"] = str(c.get("sample", ""))[:200]
        for name in ("table_variant", "level_parallel", "all_cores"):
            if isinstance(c.get(name), dict):
                out["cpu_baseline"][name + "_value"] = _r(c[name].get("value"))
                out["cpu_baseline"][name + "_cores"] = c[name].get("cores")
    if "parity_gates" in full:
        out["parity_gates"] = full["parity_gates"]
    if "extra" in full:
        ex = {}
        for name, e in full["extra"].items():
            if name == "periodic_table":
                ex[name] = {k: (_r(v) if not isinstance(v, list) else [_r(x, 4) for x in v]) for k, v in e.items()
                            if k in ("seconds", "shard_seconds", "slowest_rank", "atoms_per_rank", "atoms", "finished", "atom_steps", "mode", "measured")}
                continue
            if "ms_per_step" not in e:
                ex[name] = {k: _r(v) for k, v in e.items() if not isinstance(v, (dict, list))} or {"see": "side file"}
                continue
            k2 = e.get("kernels", {})
            ph = e.get("phase_ms_per_step", {})
            pk = k2.get("poisson", {})
            pfrac, pcnt = pk.get("frac"), pk.get("frac_counters")
            ex[name] = {"ms_per_step": _r(e["ms_per_step"], 5), "sweeps_per_s": _r(e.get("sweeps_executed_per_s"), 5), "vcycles_per_s": _r(e.get("vcycles_per_s"), 5),
                        "levels_ms": _r(ph.get("levels"), 4), "poisson_ms": _r(ph.get("poisson"), 4),
                        "issued_per_useful": _r(e.get("issued_per_useful"), 3),
                        "sweep_frac": _r(k2.get("sweep", {}).get("frac"), 3),
                        # an algorithmic fraction above 1 shows fusion, not bandwidth: never printed without the counter figure next to it
                        "poisson_frac": _r(pfrac, 3) if (pfrac is None or pfrac <= 1.0 or pcnt is not None) else "withheld (>1 by SURVEY-8d bytes, no counters)",
                        "poisson_frac_counters": _r(pcnt, 3), "poisson_frac_compulsory": _r(pk.get("frac_compulsory"), 3)}
            g = e.get("parity_gates")
            if g:
                ex[name]["gates"] = {"counts": "exact" if str(g.get("node_counts", "")).startswith("bit-exact") else "exact outside band 1.5e-10|E|+2e-9",
                                     "dE": "2e-12" if "2e-12" in str(g.get("per_level_dE")) else "6e-11|E|+6e-10", "Etot": g.get("etotal_rel"),
                                     "vcyc": _r(e.get("vcycles_per_solve"), 3)}
        out["extra"] = ex
    if full.get("full_result"):
        out["full_result"] = full["full_result"]
    line = json.dumps(out)
    if len(line) > LINE_LIMIT:                 # never lose the line to its own size: drop the optional parts, largest first
        for k in ("extra", "kernels", "phase_ms_per_step"):
            out.pop(k, None)
            line = json.dumps(out)
            if len(line) <= LINE_LIMIT:
                break
    return line


def write_full(full):
    """the complete result (every kernel object with its notes) next to the profiles and, on a gpurun box, under gpurun_out/"""
    written = []
    for d in ("profiles", "gpurun_out"):
        try:
            os.makedirs(os.path.join(ROOT, d), exist_ok=True)
            path = os.path.join(ROOT, d, "bench_full_last.json")
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
            written.append(os.path.join(d, "bench_full_last.json"))
        except OSError:
            pass
    return written


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        cpu_worker(int(sys.argv[2]), bool(int(sys.argv[3])), int(sys.argv[4]), bool(int(sys.argv[5])), int(sys.argv[6]) if len(sys.argv) > 6 else 1)
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--atoms", type=int, default=1, help="identical Rn atoms advanced together per GPU")
    ap.add_argument("--levels", type=int, default=17)
    ap.add_argument("--lsda", action="store_true")
    ap.add_argument("--tolerance", action="store_true", help="the multigrid smoother's opt-in tolerance mode (DFTA_POISSON_TOLERANCE) for the headline workload")
    ap.add_argument("--adaptive", action="store_true", help="DFTA_POISSON_ADAPTIVE for the headline workload: the tolerance kernels, V-cycles stop on the round-off floor (6-8 instead of the reference's 100)")
    ap.add_argument("--scan-sweeps", action="store_true", help="the sweeps' opt-in tolerance mode (DFTA_SWEEPS_TOLERANCE: transfer-matrix scans) for the headline workload")
    ap.add_argument("--tree-depth", type=int, default=0)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra workloads (256-atom batch, LSDA, 1 048 577 nodes)")
    ap.add_argument("--all-extras", action="store_true", help="also: LSDA in tolerance mode, a 1024-atom batch, the dense-K sweep benchmark")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="CPU baseline: skip the one-replica-per-physical-core leg")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--no-periodic-table", action="store_true", help="skip BASELINE config 4 (Z = 1..86, sharded by atom over the ranks) after the timed steps")
    ap.add_argument("--pt-zmax", type=int, default=86, help="last atom of the periodic-table sweep (tests shorten it)")
    args = ap.parse_args()

    import torch            # before dftatom_amd: one HIP runtime per process
    import torch.distributed as dist

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, torch.distributed.run
        # over 127.0.0.1) BEFORE anything touches the GPU -- torch.cuda.device_count() does not initialise HIP -- and return
        # the job's exit code.  The ranks print the one JSON line (rank 0) on this process' stdout.
        shared_test = os.environ.get("DFTA_BENCH_SHARED_GPU") == "1"
        have = torch.cuda.device_count()
        if have < args.gpus and not shared_test:
            sys.stderr.write("bench.py: --gpus %d requested but only %d HIP device(s) are visible\n" % (args.gpus, have))
            raise SystemExit(2)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d does not match WORLD_SIZE=%d of the launcher\n" % (args.gpus, world))
        raise SystemExit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    # DFTA_BENCH_SHARED_GPU=1 (testing the N > 1 control flow on a one-GPU box): every rank uses device 0 and the
    # collectives run over gloo on host tensors; never a measurement
    shared = world > 1 and os.environ.get("DFTA_BENCH_SHARED_GPU") == "1"
    dev_index = 0 if shared else local_rank
    cdev = "cpu" if shared else "cuda"
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if shared else "nccl", rank=rank, world_size=world)

    import dftatom_amd as D
    delta, rmax = GRIDS.get(args.levels, (1e-4, 50.0))
    stream = torch.cuda.current_stream().cuda_stream
    ctx = D.Context(dev_index, stream)
    grid = D.Grid(ctx, args.levels, delta, rmax)
    if rank == 0:
        try:
            HBM_MEASURED["copy"], HBM_MEASURED["triad"] = ctx.measure_hbm(1 << 27, 5)      # 3 x 1 GiB arrays, a few ms
        except Exception:
            pass
    records = torch.zeros((args.atoms, D.RECORD_DOUBLES), dtype=torch.float64, device="cuda")

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def gather(scf):
        scf.records_into(records.data_ptr())
        if world > 1:
            mine = records.to(cdev)
            gathered = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(gathered, mine)

    scf, tot = run_workload(D, ctx, grid, args.levels, args.atoms, args.lsda, args.steps, args.warmup, args.tree_depth, barrier, torch, gather,
                            poisson_mode=D.POISSON_ADAPTIVE if args.adaptive else (D.POISSON_TOLERANCE if args.tolerance else D.POISSON_EXACT),
                            sweep_mode=D.SWEEPS_TOLERANCE if args.scan_sweeps else D.SWEEPS_EXACT)
    elapsed = tot["elapsed"]
    # max over ranks of the elapsed time, sums of the work
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        w = torch.tensor([tot["sweeps_reference"], tot["sweeps_issued"], tot["vcycles"], tot["sweeps_reference_executed"]], dtype=torch.float64, device=cdev)
        dist.all_reduce(w, op=dist.ReduceOp.SUM)
        ref_all, issued_all, vc_all, exe_all = (float(x) for x in w.tolist())
    else:
        ref_all, issued_all, vc_all, exe_all = (float(tot[k]) for k in ("sweeps_reference", "sweeps_issued", "vcycles", "sweeps_reference_executed"))
    scf.close()
    pt_extra = None
    if not args.no_periodic_table and not args.no_extras and args.atoms == 1 and not args.lsda and not args.tolerance and not args.scan_sweeps and not args.adaptive:
        t_w = time.time()
        pt_extra = periodic_table(D, ctx, grid, args.levels, world, rank, dist, torch, shared, args.pt_zmax)
        if rank == 0:
            sys.stderr.write("bench.py: periodic table: %.1f s\n" % (time.time() - t_w))

    if rank == 0:
        ncu, devname = ctx.device_info()
        pmode = "adaptive" if args.adaptive else ("tolerance" if args.tolerance else "exact")
        wl = workload_tag(args.levels, args.atoms, args.lsda, pmode, "tolerance" if args.scan_sweeps else "exact")
        sweep, pois = kernel_figures(tot, args.levels, grid.N, args.atoms, wl)
        dominant = pois if tot["ms_poisson"] >= tot["ms_sweep_kernels"] else sweep
        roof = {"bound": "hbm", "kernel": dominant["kernel"],
                "achieved": dominant["algorithmic_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dominant["frac"],
                "achieved_is": "algorithmic bytes of SURVEY 8d per launch / average launch duration (HIP events on the launch stream)",
                "traffic": dominant.get("hbm_bytes_per_launch_counters"), "traffic_source": dominant.get("counters_profile"),
                "hbm_GBps_counters": dominant.get("hbm_GBps_counters"), "frac_counters": dominant.get("frac_counters"),
                "bytes_per_launch": dominant["bytes_per_launch"],
                "avg_launch_ms": dominant["avg_launch_ms"], "launches": dominant["launches"],
                # the same kernel's average in the committed rocprofv3 kernel trace of this workload, and the launches it averages
                "avg_launch_ms_rocprof": dominant.get("avg_launch_ms_rocprof"), "rocprof_window": dominant.get("rocprof_window"),
                "peak_measured_copy": HBM_MEASURED["copy"], "peak_measured_triad": HBM_MEASURED["triad"],
                "share_of_step_ms": {"multigrid kernel": tot["ms_poisson"] / args.steps, "sweep kernel": tot["ms_sweep_kernels"] / args.steps},
                "binding_resource": dominant["binding_resource"],
                "note": "dominant kernel of the timed region by HIP-event time; see `kernels` for both hot kernels (the sweep kernel also on "
                        "issued bytes and against its VALU-issue ceiling)"}
        if dominant is sweep:
            roof["frac_issued"] = sweep["frac_issued"]
        out = {
            "metric": "numerov_sweeps_per_s (executed sweeps of the reference's bisection path, whole SCF step; Rn Z=86 @ %d pts)" % grid.N,
            "value": exe_all / elapsed,
            "value_reference_equivalent": ref_all / elapsed,
            "unit": "sweeps/s",
            "n_gpus": world,
            "ranks_seen": (dist.get_world_size() if world > 1 and dist.is_initialized() else 1),     # what the process group really has
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (reference's flat start density, SCF iterations %d..%d)" % (args.warmup, args.warmup + args.steps - 1),
            "config": {"workload": "Rn Z=86 %s, %d levels (%d pts), delta=%g, Rmax=%g, mixing 0.5, %d atom(s)/GPU, %s"
                                   % ("LSDA" if args.lsda else "LDA", args.levels, grid.N, delta, rmax, args.atoms, tot["levels_layout"]),
                       "atoms_per_gpu": args.atoms, "parallelism": "replicas x%d" % world,
                       "poisson_mode": pmode, "sweep_mode": "tolerance (scan)" if args.scan_sweeps else "exact"},
            "scf_step_ms": 1e3 * elapsed / args.steps,
            "value_definition": "value = sweeps on the reference's bisection path that are actually integrated here (CountNodes + SolutionInZero + "
                                "Match) / whole-step wall time; value_reference_equivalent also counts the ~52 CountNodes calls per node-less level's "
                                "second bisection that the reference integrates and this path decides without integrating (+9 %)",
            "sweeps_executed_per_s": exe_all / elapsed,
            "sweeps_issued_per_s": issued_all / elapsed,
            "poisson_vcycles_per_s": vc_all / elapsed,
            "poisson_vcycles_per_s_kernel": pois["vcycles_per_s"],
            "phase_ms_per_step": {"levels": tot["ms_levels"] / args.steps, "sweep_kernels": tot["ms_sweep_kernels"] / args.steps,
                                  "poisson": tot["ms_poisson"] / args.steps,
                                  "tail": tot["ms_tail"] / args.steps, "hip_event_total": tot["ev_ms"] / args.steps},
            "rounds_per_step": tot["rounds"] / args.steps,
            "energies_last_step": tot["energies"],
            "device": devname, "compute_units": ncu,
            "roofline": roof,
            "kernels": {"sweep": sweep, "poisson": pois},
            "parity_gates": parity_gates(pmode, "tolerance" if args.scan_sweeps else "exact"),
        }
        cpu_job = None
        want_cpu = world == 1 and not args.no_cpu      # rank 0 at N = 1 only: the other ranks of a larger job would sit at the final barrier
        # The CPU legs run in child processes on the host cores NEXT TO the long extras (batches, 2^20+1 nodes) -- but start only after the
        # short single-atom extras: three Python children importing numpy while a 16 ms step is being timed cost it 3-5 ms of host jitter
        if want_cpu and args.no_extras:
            cpu_job = cpu_baseline_start(args.levels, args.lsda, args.cpu_steps)
        if world == 1 and not args.no_extras:
            # further measured workloads with the same per-kernel figures (>= 10 timed steps after >= 5 warm-up steps wherever a step
            # is short enough): the opt-in tolerance mode of the multigrid smoother, LSDA (BASELINE config 3), a machine-filling
            # batch of Rn atoms, and the 1 048 577-node stress of config 5 (one atom and a batch of 16, SCF steps 6.. : from the seventh step
            # on one level ends its third bisection at the reference's 500-iteration cap in most steps); a few seconds each.
            # --all-extras adds LSDA in tolerance mode, the 1024-atom batch and the dense-K sweep benchmark of SURVEY 8d.
            extra = {}
            TOL, SCAN, ADAPT = D.POISSON_TOLERANCE, D.SWEEPS_TOLERANCE, D.POISSON_ADAPTIVE
            # default: one object per BASELINE config that fits one GPU (configs[2] Rn LSDA, configs[4] 1 048 577 nodes x 1 and x 16) and the
            # opt-in modes of the headline workload; everything else behind --all-extras (the default run stays within two minutes)
            sel = [("rn_lda_both_tolerance_modes", args.levels, 1, False, 10, 5, TOL, SCAN, "scan_tol"),
                   # DFTA_POISSON_ADAPTIVE: the V-cycles stop on the round-off floor (6 .. 8 per solve) instead of at the reference's cap of 100
                   ("rn_lda_scan_sweeps_adaptive_vcycles", args.levels, 1, False, 10, 5, ADAPT, SCAN, "scan_adaptive"),
                   ("rn_lsda", args.levels, 1, True, 10, 5, None, None, "rn_lsda"),
                   ("batch256_lda", args.levels, 256, False, 6, 5, None, None, "batch256"),
                   # the throughput workload in the opt-in modes: the scan sweeps of 256 atoms stream 3.3 - 6.7 TB/s of table rows (profiles/*_batch256_scan_tol_*)
                   ("batch256_lda_both_tolerance_modes", args.levels, 256, False, 6, 5, TOL, SCAN, "batch256_scan_tol"),
                   ("rn_lsda_l20", 20, 1, True, 6, 6, None, None, "l20"),
                   ("rn_lsda_l20_batch16", 20, 16, True, 4, 6, None, None, "l20_batch16")]
            if args.all_extras:
                sel += [("rn_lda_scan_sweeps", args.levels, 1, False, 10, 5, None, SCAN, "scan"),
                        ("rn_lda_poisson_tolerance", args.levels, 1, False, 10, 5, TOL, None, "tolerance"),
                        ("rn_lsda_both_tolerance_modes", args.levels, 1, True, 10, 5, TOL, SCAN, None),
                        ("batch256_lda_scan_sweeps", args.levels, 256, False, 6, 5, None, SCAN, "batch256_scan"),
                        ("batch256_lda_scan_sweeps_adaptive_vcycles", args.levels, 256, False, 6, 5, ADAPT, SCAN, None),
                        ("rn_lsda_l20_scan_sweeps", 20, 1, True, 6, 6, None, SCAN, None),
                        ("rn_lsda_l20_both_tolerance_modes", 20, 1, True, 6, 6, TOL, SCAN, "l20_scan_tol"),
                        ("rn_lsda_l20_batch16_both_tolerance_modes", 20, 16, True, 4, 6, TOL, SCAN, "l20_batch16_scan_tol"),
                        ("rn_lsda_poisson_tolerance", args.levels, 1, True, 10, 5, TOL, None, None),
                        ("batch1024_lda", args.levels, 1024, False, 4, 2, None, None, None),
                        ("batch1024_lda_scan_sweeps", args.levels, 1024, False, 4, 2, None, SCAN, None)]
            for name, lv, atoms, lsda, st, wu, pm, sm, wl2 in sel:
                if want_cpu and cpu_job is None and (atoms > 1 or lv != args.levels):
                    cpu_job = cpu_baseline_start(args.levels, args.lsda, args.cpu_steps)
                if lv == args.levels:
                    g2, d2, r2 = grid, delta, rmax
                else:
                    d2, r2 = GRIDS[lv]
                    g2 = D.Grid(ctx, lv, d2, r2)
                t_w = time.time()
                s2, t2 = run_workload(D, ctx, g2, lv, atoms, lsda, st, wu, 0, barrier, torch, poisson_mode=pm, sweep_mode=sm)
                s2.close()
                sys.stderr.write("bench.py: extra %s: %.1f s\n" % (name, time.time() - t_w))
                pm_name = {D.POISSON_TOLERANCE: "tolerance", D.POISSON_ADAPTIVE: "adaptive"}.get(pm, "exact")
                extra[name] = summarize(t2, lv, g2.N, atoms, lsda, world, d2, r2, workload_tag(lv, atoms, lsda, pm_name, "tolerance" if sm == D.SWEEPS_TOLERANCE else "exact"))
                extra[name]["poisson_mode"] = {D.POISSON_TOLERANCE: "tolerance", D.POISSON_ADAPTIVE: "adaptive (tolerance kernels, V-cycles stop on the round-off floor)"}.get(pm, "exact")
                extra[name]["sweep_mode"] = "tolerance (scan)" if sm == D.SWEEPS_TOLERANCE else "exact"
                extra[name]["parity_gates"] = parity_gates({D.POISSON_TOLERANCE: "tolerance", D.POISSON_ADAPTIVE: "adaptive"}.get(pm, "exact"),
                                                           "tolerance" if sm == D.SWEEPS_TOLERANCE else "exact")
                extra[name]["vcycles_per_solve"] = t2["vcycles"] / max(t2["steps"] * atoms, 1)
                extra[name]["warmup"] = wu
                if g2 is not grid:
                    g2.close()
            if args.all_extras:
                try:
                    sys.path.insert(0, os.path.join(ROOT, "profiles"))
                    import dense_k_sweep
                    extra["dense_k_sweeps"] = dense_k_sweep.run(D, ctx, grid, HBM_PEAK_GBS, HBM_MEASURED["copy"])
                except Exception as e:                      # the isolated sweep-kernel benchmark must not cost the line
                    extra["dense_k_sweeps"] = {"error": repr(e)}
            out["extra"] = extra
        if want_cpu and cpu_job is None:
            cpu_job = cpu_baseline_start(args.levels, args.lsda, args.cpu_steps)
        if pt_extra is not None:
            out.setdefault("extra", {})["periodic_table"] = pt_extra
        if world == 1 and not args.no_extras:
            pcs = per_call_surface()
            if pcs is not None:
                out.setdefault("extra", {})["per_call_surface"] = pcs
        if cpu_job is not None:
            t_w = time.time()
            out["cpu_baseline"] = cpu_baseline_finish(cpu_job, not args.no_cpu_all_cores)
            sys.stderr.write("bench.py: waiting for the CPU legs (+ all-core replicas): %.1f s\n" % (time.time() - t_w))
        if shared:
            out["data"] += " -- SHARED-GPU TEST MODE (all ranks on device 0, gloo): not a measurement"
        out["full_result"] = write_full(out)
        sys.stdout.flush()
        print(compact_line(out))
        sys.stdout.flush()
    grid.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
