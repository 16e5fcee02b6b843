"""CPU suite: bench.py's stdout line is built by a pure function (compact_line) -- it must stay below the 8 KB the driver keeps of
stdout (round 3's 31.5 KB line was cut and parsed as null), round-trip through json and carry the contract's objects."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def canned_tot(atoms=1, steps=20):
    return {"sweeps_issued": 1175327 * steps // 16, "sweeps_reference": 2270 * steps, "sweeps_reference_executed": 2066 * steps,
            "points_traversed": 9.1e9 * steps, "points_reference": 2.2e8 * steps, "vcycles": 100 * steps * atoms, "rounds": int(7.25 * steps),
            "ms_sweep_kernels": 29.4 * steps, "ms_levels": 34.05 * steps, "ms_poisson": 27.7 * steps, "ms_tail": 0.5 * steps,
            "elapsed": 0.0623 * steps, "ev_ms": 62.3 * steps, "steps": steps, "trials_per_round": 16384, "tree_depth": 10,
            "levels_layout": "latency mode (slots re-allotted every round)", "poisson_G": 33,
            "energies": [-21861.346869, 21861.3, -8632.0, -53000.0, -387.4]}


def full_result(bench):
    tot = canned_tot()
    sweep, pois = bench.kernel_figures(tot, 17, 131073, 1, "default")
    dominant = sweep
    full = {"metric": "numerov_sweeps_per_s (executed sweeps of the reference's bisection path, whole SCF step; Rn Z=86 @ 131073 pts)",
            "value": 33148.123456789, "value_reference_equivalent": 36300.5, "unit": "sweeps/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 62.3123456, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (reference's flat start density, SCF iterations 5..24)",
            "config": {"workload": "Rn Z=86 LDA, 17 levels (131073 pts), delta=0.0001, Rmax=50, mixing 0.5, 1 atom(s)/GPU, latency mode",
                       "atoms_per_gpu": 1, "parallelism": "replicas x1", "poisson_mode": "exact"},
            "sweeps_issued_per_s": 1175327.0, "poisson_vcycles_per_s": 1604.0, "poisson_vcycles_per_s_kernel": 3610.0,
            "phase_ms_per_step": {"levels": 34.05, "sweep_kernels": 29.4, "poisson": 27.7, "tail": 0.5, "hip_event_total": 62.3},
            "rounds_per_step": 7.25, "energies_last_step": tot["energies"], "device": "AMD Instinct MI355X", "compute_units": 256,
            "roofline": {"bound": "hbm", "kernel": dominant["kernel"], "achieved": dominant["algorithmic_GBps"], "peak": 8000.0, "unit": "GB/s",
                         "frac": dominant["frac"], "frac_issued": dominant["frac_issued"], "traffic": 53.6e6, "bytes_per_launch": dominant["bytes_per_launch"],
                         "avg_launch_ms": dominant["avg_launch_ms"], "launches": dominant["launches"], "hbm_GBps_counters": 13.2, "frac_counters": 0.0016,
                         "binding_resource": "x" * 300, "note": "y" * 400},
            "kernels": {"sweep": sweep, "poisson": pois},
            "cpu_baseline": {"value": 573.6, "unit": "sweeps/s", "cores": 1, "kind": "port", "sample": "3 SCF steps of Rn LDA @ 17 levels " + "z" * 300,
                             "ms_per_step": 3736.8, "vcycles_per_s": 26.7, "cpu_model": "AMD EPYC 9575F 64-Core Processor", "nproc": 256,
                             "table_variant": {"value": 929.0, "cores": 1, "note": "n" * 200},
                             "level_parallel": {"value": 2075.0, "cores": 15, "note": "n" * 300},
                             "all_cores": {"value": 6417.0, "cores": 128, "note": "n" * 500}},
            "extra": {}}
    for name, atoms in (("rn_lda_scan_sweeps", 1), ("rn_lda_both_tolerance_modes", 1), ("rn_lda_poisson_tolerance", 1), ("rn_lsda", 1),
                        ("rn_lsda_both_tolerance_modes", 1), ("batch256_lda", 256), ("batch256_lda_scan_sweeps", 256), ("batch1024_lda", 1024),
                        ("batch1024_lda_scan_sweeps", 1024), ("rn_lsda_l20", 1), ("rn_lsda_l20_scan_sweeps", 1), ("rn_lsda_l20_batch16", 16)):
        full["extra"][name] = bench.summarize(canned_tot(atoms, 10), 17, 131073, atoms, False, 1, 1e-4, 50.0, None)
        full["extra"][name]["warmup"] = 5
    full["extra"]["dense_k_sweeps"] = {"K512_count_nodes": {"sweeps": 7680, "note": "q" * 2000}, "error": None}
    full["parity_gates"] = bench.parity_gates("exact", "exact")
    full["extra"]["rn_lda_both_tolerance_modes"]["parity_gates"] = bench.parity_gates("tolerance", "tolerance")
    full["extra"]["rn_lda_both_tolerance_modes"]["vcycles_per_solve"] = 100.0
    full["extra"]["periodic_table"] = {"workload": "w" * 300, "seconds": 19.0123456, "shard_seconds": [19.0123456], "slowest_rank": 0, "atoms_per_rank": [86],
                                       "atoms": 86, "finished": 61, "atom_steps": 5721, "mode": "exact kernels (default path)", "measured": "this run, 1 GPU(s)"}
    return full


def test_line_is_compact_and_complete():
    import bench
    full = full_result(bench)
    assert len(json.dumps(full)) > 20000                      # the full result is what used to be printed
    line = bench.compact_line(full)
    assert len(line) < 7000 and "\n" not in line
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["workload"].startswith("Rn Z=86 LDA") and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "bytes_per_launch", "avg_launch_ms", "launches"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 * r["frac"] + 1e-12
    c = d["cpu_baseline"]
    assert c["value"] == 573.6 and c["cores"] == 1 and c["kind"] == "port" and c["unit"] == "sweeps/s" and len(c["sample"]) <= 200
    assert c["level_parallel_value"] == 2075.0 and c["all_cores_cores"] == 128
    assert set(d["extra"]) >= {"rn_lsda", "batch256_lda", "rn_lsda_l20_batch16", "rn_lda_scan_sweeps", "rn_lda_both_tolerance_modes"}
    for name, e in d["extra"].items():
        assert all(not isinstance(v, (dict, list)) for k, v in e.items() if k not in ("gates", "shard_seconds", "atoms_per_rank")), name     # flat
    assert d["extra"]["rn_lsda"]["ms_per_step"] > 0 and d["extra"]["rn_lsda"]["poisson_frac"] > 0
    # what kind of number a line is: the gates of the mode it was measured in travel with it
    assert d["parity_gates"]["vcycles_per_solve"] == 100 and d["parity_gates"]["etotal_rel"] == 1e-9 and d["parity_gates"]["node_counts"].startswith("bit-exact")
    g = d["extra"]["rn_lda_both_tolerance_modes"]["gates"]
    assert g["counts"].startswith("exact outside band") and g["dE"] == "6e-11|E|+6e-10" and g["Etot"] == 2e-9 and g["vcyc"] == 100.0
    # BASELINE config 4 on the line: seconds of the whole sweep, the ranks' shard times, atoms per rank
    pt = d["extra"]["periodic_table"]
    assert pt["seconds"] == 19.0123 and pt["atoms_per_rank"] == [86] and pt["finished"] == 61 and "workload" not in pt


def test_fraction_above_one_is_withheld_without_counters():
    """batch256's multigrid reads 1.01 of the roofline in SURVEY-8d bytes (every pass counted) -- fusion, not bandwidth.
    The line prints such a figure only with the counter-based fraction next to it; the compulsory-traffic fraction (24 N per V-cycle) is
    always there."""
    import bench
    full = full_result(bench)
    e = full["extra"]["batch256_lda"]
    ratio = e["kernels"]["poisson"]["frac_compulsory"] / e["kernels"]["poisson"]["frac"]
    assert abs(ratio - 24 * 131073 / 49285736) < 1e-12                  # 24 N of the 376 N of SURVEY 8d
    e["kernels"]["poisson"]["frac"] = 1.01
    e["kernels"]["poisson"]["frac_counters"] = None
    d = json.loads(bench.compact_line(full))
    assert isinstance(d["extra"]["batch256_lda"]["poisson_frac"], str) and "withheld" in d["extra"]["batch256_lda"]["poisson_frac"]
    assert 0 < d["extra"]["batch256_lda"]["poisson_frac_compulsory"] < 1
    e["kernels"]["poisson"]["frac_counters"] = 0.495
    d = json.loads(bench.compact_line(full))
    assert d["extra"]["batch256_lda"]["poisson_frac"] == 1.01 and d["extra"]["batch256_lda"]["poisson_frac_counters"] == 0.495


def test_line_survives_oversized_optional_parts():
    """whatever the extras grow into, the schema keys + roofline + cpu_baseline stay on a line the driver can keep"""
    import bench
    full = full_result(bench)
    for i in range(400):
        full["extra"]["w%d" % i] = dict(full["extra"]["rn_lsda"])
    line = bench.compact_line(full)
    assert len(line) < 8000
    d = json.loads(line)
    assert "roofline" in d and "cpu_baseline" in d and d["value"] > 0 and "extra" not in d


def test_algorithmic_figures_are_not_called_bandwidth():
    """a 'frac' above 1 against a measured copy bandwidth is no evidence; the kernel objects name the SURVEY-8d figure
    'algorithmic' and carry the counter-based rate separately"""
    import bench
    sweep, pois = bench.kernel_figures(canned_tot(256, 10), 17, 131073, 256, None)
    for k in (sweep, pois):
        assert "algorithmic_GBps" in k and "achieved" not in k and "frac_measured" not in k


def test_rocprof_average_is_read_from_the_committed_summary(tmp_path, monkeypatch):
    """the line carries the profiler's average launch duration of the dominant kernel over the TIMED launches of the
    committed kernel trace of the same workload, with the window it describes"""
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r98_default_bench_kernel_stats.csv").write_text(
        "kernel,calls,total_ms,avg_us,min_us,max_us,percent,timed_calls,timed_avg_us,timed_min_us,timed_max_us\n"
        "k_levels_persist,25,770.0,30800.0,27000.0,43000.0,53.0,20,29400.0,27000.0,33000.0\n"
        "k_expand,300,3.0,10.0,8.0,12.0,0.2,,,,\n")
    (prof / "r98_default_bench_under_rocprof.json").write_text(json.dumps({"steps": 20, "warmup": 5}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    ms, win = bench.rocprof_launch("k_levels_persist", "default")
    assert ms == 29.4 and win["launches"] == 20 and "steps 5..24" in win["window"] and win["profile"].startswith("r98_default")
    ms, win = bench.rocprof_launch("k_expand", "default")          # many launches per step: no timed window, all launches
    assert ms == 0.01 and "ALL launches" in win["window"]
    assert bench.rocprof_launch("k_nothing", "default") == (None, None)
    assert bench.rocprof_launch("k_levels_persist", "no_such_workload") == (None, None)


def test_committed_line_agrees_with_the_committed_rocprof_summary():
    """the newest committed bench line of the default workload (profiles/r??_bench_default_line.json) against the rocprofv3 kernel trace
    committed with it: the HIP-event average of the dominant kernel and the profiler's average over the same window differ by < 5 %,
    and `frac` follows from the CSV alone: bytes_per_launch / timed_avg / peak"""
    import csv
    import glob
    lines = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default_line.json")))
    d = json.load(open(lines[-1]))
    r = d["roofline"]
    if "avg_launch_ms_rocprof" not in r:
        import pytest
        pytest.skip("the newest committed line predates roofline.avg_launch_ms_rocprof (%s)" % os.path.basename(lines[-1]))
    assert r["avg_launch_ms_rocprof"] is not None, "the line was printed without a committed kernel trace of its workload"
    assert abs(r["avg_launch_ms_rocprof"] - r["avg_launch_ms"]) <= 0.05 * r["avg_launch_ms"], (r["avg_launch_ms_rocprof"], r["avg_launch_ms"])
    assert "timed launches only" in r["rocprof_window"] and "--steps %d --warmup %d" % (d["steps"], d["warmup"]) in r["rocprof_window"]
    prof = r["rocprof_window"].split(":", 1)[0]
    rows = {x["kernel"]: x for x in csv.DictReader(open(os.path.join(ROOT, "profiles", prof)))}
    t_ms = float(rows[r["kernel"]]["timed_avg_us"]) / 1e3
    frac_from_csv = r["bytes_per_launch"] / (t_ms * 1e-3) / 1e9 / r["peak"]
    assert abs(frac_from_csv - r["frac"]) <= 0.05 * r["frac"], (frac_from_csv, r["frac"])
