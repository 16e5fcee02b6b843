"""Turn the raw rocprofv3 output of profiles/collect.sh into the small summaries that are committed:
    profiles/<tag>_bench_kernel_stats.csv   per-kernel calls / total / average / min / max duration (kernel trace); timed_*: the same over
                                            the launches of the TIMED steps only (kernels launched once per step; warm-up skipped by index)
    profiles/<tag>_bench_under_rocprof.json the bench.py line printed by the profiled run
    profiles/<tag>_hbm_traffic.json         FETCH_SIZE / WRITE_SIZE per kernel and launch (separate PMC passes)
    profiles/<tag>_sq_counters.json         SQ counters per kernel and launch
Usage: python3 profiles/summarize.py gpurun_out/<tag> <tag> "<bench args>"
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, pattern):
    return sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))


def short(name):
    m = re.search(r"(k_[A-Za-z0-9_]+)", name)
    return m.group(1) if m else name[:60]


def window_of(args):
    """(warmup, steps) of the profiled bench command"""
    m_s, m_w = re.search(r"--steps\s+(\d+)", args), re.search(r"--warmup\s+(\d+)", args)
    return (int(m_w.group(1)) if m_w else 2), (int(m_s.group(1)) if m_s else 5)


def kernel_stats(d, warmup=None, steps=None):
    """per kernel: all launches, and -- for a kernel launched exactly once per SCF step (warmup + steps launches, one more for the multigrid's set-up
    solve: the device-side level search, the multigrid solve) -- the TIMED launches alone: the last `steps` ones in start order, i.e. the window bench.py's line is
    measured on (the warm-up launches are skipped by index)"""
    files = find(d, "*kernel_trace.csv")
    agg = defaultdict(list)
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                agg[short(row["Kernel_Name"])].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    rows = []
    total = sum(sum(x[1] for x in v) for v in agg.values()) or 1
    for k, v in sorted(agg.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
        v.sort()
        dur = [x[1] for x in v]
        r = {"kernel": k, "calls": len(dur), "total_ms": sum(dur) / 1e6, "avg_us": sum(dur) / len(dur) / 1e3,
             "min_us": min(dur) / 1e3, "max_us": max(dur) / 1e3, "percent": 100.0 * sum(dur) / total,
             "timed_calls": "", "timed_avg_us": "", "timed_min_us": "", "timed_max_us": ""}
        # (+ 1: the solve of the flat start density when the SCF object is made)
        if warmup is not None and steps and len(dur) in (warmup + steps, warmup + steps + 1):
            t = dur[-steps:]
            r.update({"timed_calls": len(t), "timed_avg_us": sum(t) / len(t) / 1e3, "timed_min_us": min(t) / 1e3, "timed_max_us": max(t) / 1e3})
        rows.append(r)
    return rows


def counters(d):
    """{kernel: {counter: [value per dispatch]}}"""
    out = defaultdict(lambda: defaultdict(list))
    for f in find(d, "*counter_collection.csv"):
        with open(f) as fh:
            per_dispatch = defaultdict(lambda: defaultdict(float))
            names = {}
            for row in csv.DictReader(fh):
                did = row.get("Dispatch_Id") or row.get("Correlation_Id")
                names[did] = short(row["Kernel_Name"])
                per_dispatch[did][row["Counter_Name"]] += float(row["Counter_Value"])
            for did, cs in per_dispatch.items():
                for c, v in cs.items():
                    out[names[did]][c].append(v)
    return out


def main():
    raw, tag = sys.argv[1], sys.argv[2]
    args = sys.argv[3] if len(sys.argv) > 3 else ""
    pdir = os.path.join(ROOT, "profiles")
    warmup, steps = window_of(args)
    rows = kernel_stats(os.path.join(raw, "trace"), warmup, steps)
    if rows:
        with open(os.path.join(pdir, tag + "_bench_kernel_stats.csv"), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
            w.writeheader()
            for r in rows:
                w.writerow({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()})
    bj = os.path.join(raw, "bench_under_rocprof.json")
    if os.path.exists(bj):
        lines = [ln for ln in open(bj).read().splitlines() if ln.startswith("{")]
        if lines:
            with open(os.path.join(pdir, tag + "_bench_under_rocprof.json"), "w") as fh:
                json.dump(json.loads(lines[-1]), fh, indent=1)
    fetch, write = counters(os.path.join(raw, "fetch")), counters(os.path.join(raw, "write"))
    sys.path.insert(0, ROOT)
    import bench
    traffic = {"source_sha": bench.source_sha(),
               "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> --kernel-trace --output-format csv -- python3 bench.py " + args +
                          "  (separate passes, MI355X gfx950, ROCm 7.2; profiles/collect.sh)",
               "unit": "FETCH_SIZE / WRITE_SIZE as reported by rocprofv3 (KiB); bytes = value * 1024.  MI355X_MICROARCH.md (HBM): on gfx950 "
                       "FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads and is uncalibrated for other access "
                       "widths; WRITE_SIZE is exact for streaming stores.  Both the raw sum and the sum with FETCH_SIZE doubled are given.",
               "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        e = {}
        for name, src in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
            v = src.get(k, {}).get(name, [])
            if v:
                e[name] = {"launches": len(v), "mean_KiB_per_launch": sum(v) / len(v), "max_KiB": max(v)}
        f = e.get("FETCH_SIZE", {}).get("mean_KiB_per_launch", 0.0)
        w = e.get("WRITE_SIZE", {}).get("mean_KiB_per_launch", 0.0)
        e["hbm_bytes_per_launch_raw"] = (f + w) * 1024
        e["hbm_bytes_per_launch_fetch_doubled"] = (2 * f + w) * 1024
        traffic["kernels"][k] = e
    if traffic["kernels"]:
        with open(os.path.join(pdir, tag + "_hbm_traffic.json"), "w") as fh:
            json.dump(traffic, fh, indent=1)
    sq = counters(os.path.join(raw, "sq"))
    if sq:
        o = {"source_sha": bench.source_sha(), "command": "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY "
                        "SQ_WAIT_INST_ANY --kernel-trace -- python3 bench.py " + args, "kernels": {}}
        for k, cs in sq.items():
            o["kernels"][k] = {c: {"launches": len(v), "mean_per_launch": sum(v) / len(v)} for c, v in cs.items()}
        with open(os.path.join(pdir, tag + "_sq_counters.json"), "w") as fh:
            json.dump(o, fh, indent=1)
    print("summaries written to", pdir)


if __name__ == "__main__":
    main()
