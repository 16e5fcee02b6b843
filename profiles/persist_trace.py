"""Timeline of the device-side exact level search (csrc/persist.inc) over the timed steps of the headline workload.

Runs Rn LDA @ 131 073 nodes (BASELINE configs[1]) for --steps SCF steps in a child process with DFTA_DEBUG=LEVELS_PERSIST_TRACE (the
library prints one record per closed round of every level: time since the kernel's first plan, level, round, what comes next), parses the
log and writes ONE json: per step the kernel time and, per level, the number of rounds, the time its search ended and the time its
wavefunction was matched and normalised; and over the steady-state steps (>= --skip) the figures DESIGN.md section 4.2 quotes: which level ends
last, how many dependent rounds it needed, how long one of its rounds is, and the resulting floor "rounds x round time" of the design.

    python3 profiles/persist_trace.py --out profiles/r05_levels_persist_trace.json        (on the MI355X box)
"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, %r)
import dftatom_amd as D
ctx = D.Context(0)
grid = D.Grid(ctx, 17, 1e-4, 50.0)
scf = D.Scf(ctx, grid, [86], lsda=%s)
names = scf.level_names() if hasattr(scf, "level_names") else None
for k in range(%d):
    sys.stderr.write("=== step %%d\n" %% k)
    st = scf.step()
    sys.stderr.write("=== stats levels_ms %%.3f sweep_ms %%.3f layout %%d fallbacks %%d\n" %% (st.ms_levels, st.ms_sweep_kernels, st.levels_layout, st.levels_fallbacks))
"""


def parse(log):
    steps, cur = [], None
    for line in log.splitlines():
        m = re.match(r"=== step (\d+)", line)
        if m:
            cur = {"step": int(m.group(1)), "kernel_ms": None, "levels": {}}
            steps.append(cur)
            continue
        if cur is None:
            continue
        m = re.match(r"=== stats levels_ms ([\d.]+) sweep_ms ([\d.]+) layout (\d+) fallbacks (\d+)", line)
        if m:
            cur["levels_ms"], cur["sweep_kernels_ms"], cur["layout"], cur["fallbacks"] = float(m.group(1)), float(m.group(2)), int(m.group(3)), int(m.group(4))
            continue
        m = re.match(r"persist trace: (\d+) records, kernel ([\d.]+) ms", line)
        if m:
            cur["kernel_ms"] = float(m.group(2))
            continue
        m = re.match(r"\s+([\d.]+) us  job\s+(\d+) round\s+(\d+)\s+(.*)", line)
        if not m:
            continue
        t, j, r, rest = float(m.group(1)), m.group(2), int(m.group(3)), m.group(4)
        d = cur["levels"].setdefault(j, {"rounds": 0, "round_end_us": []})
        d["rounds"] = max(d["rounds"], r)
        if r > 0:
            d["round_end_us"].append(t)
        if rest.startswith("search ended"):
            d["search_end_us"] = t
        if rest.startswith("DONE"):
            d["done_us"] = t
    return steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--skip", type=int, default=5)
    ap.add_argument("--lsda", action="store_true")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_levels_persist_trace.json"))
    a = ap.parse_args()
    env = dict(os.environ)
    env["DFTA_DEBUG"] = ",".join(x for x in (env.get("DFTA_DEBUG", ""), "LEVELS_PERSIST_TRACE") if x)
    p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, "True" if a.lsda else "False", a.steps)], env=env, capture_output=True, text=True, timeout=1200)
    if p.returncode != 0:
        sys.stderr.write(p.stderr[-4000:])
        sys.exit(p.returncode)
    steps = parse(p.stderr)
    steady = [s for s in steps if s["step"] >= a.skip and s["levels"]]
    summary = {}
    if steady:
        crit, crit_rounds, crit_round_ms, ends, dones = {}, [], [], [], []
        for s in steady:
            last = max(s["levels"].items(), key=lambda kv: kv[1].get("search_end_us", 0.0))
            crit[last[0]] = crit.get(last[0], 0) + 1
            crit_rounds.append(last[1]["rounds"])
            crit_round_ms.append(last[1].get("search_end_us", 0.0) / max(last[1]["rounds"], 1) / 1e3)
            ends.append(last[1].get("search_end_us", 0.0) / 1e3)
            dones.append(max(d.get("done_us", 0.0) for d in s["levels"].values()) / 1e3)
        mean = lambda v: sum(v) / len(v)
        summary = {"steps": len(steady), "kernel_ms_mean": mean([s["kernel_ms"] for s in steady]),
                   "last_search_end_ms_mean": mean(ends), "last_match_done_ms_mean": mean(dones),
                   "level_that_ends_last": crit, "its_rounds_mean": mean(crit_rounds), "its_rounds_min": min(crit_rounds),
                   "its_rounds_hist": {str(r): crit_rounds.count(r) for r in sorted(set(crit_rounds))},
                   "its_round_ms_mean": mean(crit_round_ms),
                   "its_rounds_mode": max(set(crit_rounds), key=crit_rounds.count),
                   "ms_mode_rounds_x_round": max(set(crit_rounds), key=crit_rounds.count) * mean(crit_round_ms),
                   "ms_min_rounds_x_round": min(crit_rounds) * mean(crit_round_ms),
                   "note": "level index = position in the batch's level list (Rn LDA: 0 = 1s, 1 = 2s, 2 = 2p, ...); a round of the level that ends last is one "
                           "full-length sweep of ~118-131 k dependent fp64 steps; 'rounds x round' = what the kernel would take with nothing but the dependent sweeps of that level on its path (mode: the usual round count; min: the steps with the longest predicted spines)"}
    out = {"workload": "Rn %s @ 131 073 nodes, one atom, exact kernels, DFTA_DEBUG=LEVELS_PERSIST_TRACE (the trace adds ~1 %% to the kernel)" % ("LSDA" if a.lsda else "LDA"),
           "summary_steady_state": summary,
           "steps": [{"step": s["step"], "kernel_ms": s["kernel_ms"], "levels_ms": s.get("levels_ms"), "layout": s.get("layout"), "fallbacks": s.get("fallbacks"),
                      "levels": {j: {"rounds": d["rounds"], "search_end_ms": round(d.get("search_end_us", 0.0) / 1e3, 3), "done_ms": round(d.get("done_us", 0.0) / 1e3, 3)}
                                 for j, d in sorted(s["levels"].items(), key=lambda kv: int(kv[0]))}} for s in steps]}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
