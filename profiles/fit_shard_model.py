"""Re-fit the shard time model of dftatom_amd/sweep.py -- T(shard) = FLOOR_SMALL_MS x (steps with <= 7 live atoms) + FLOOR_BIG_MS x (steps with
more) + JOB_MS x sum(subshells x expected steps) -- to the recorded shards of the emulated 1-, 2-, 4- and 8-rank periodic-table sweeps (examples/periodic_table.py --emulate-ranks N,
one GPU; profiles/<round>_periodic_table_predicted_scaling_<mode>.json, the newest round), by least squares, per mode of the sweeps.

    python profiles/fit_shard_model.py            # prints the pairs to paste into sweep.SHARD_MODEL and the residuals
    python profiles/fit_shard_model.py --merge gpurun_out/r06_pt_exact_{1,2,4,8}.json --mode exact --tag r06    # build the recorded file from raw runs
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def path_of(mode, tag=None):
    """the recorded emulation of `mode`: profiles/<tag>_periodic_table_predicted_scaling_<mode>.json, tag = the newest round by default"""
    import glob
    if tag is None:
        have = sorted(glob.glob(os.path.join(ROOT, "profiles", "r??_periodic_table_predicted_scaling_%s.json" % mode)))
        tag = os.path.basename(have[-1]).split("_")[0] if have else "r04"
    return os.path.join(ROOT, "profiles", "%s_periodic_table_predicted_scaling_%s.json" % (tag, mode))


def merge(files, mode, tag=None):
    runs = [json.load(open(f)) for f in files]
    runs.sort(key=lambda r: r["emulated_ranks"])
    out = {"what": "periodic table Z = 1..86 at 131 073 nodes on ONE MI355X as the shards of an N-rank sweep, each shard run alone "
                   "(examples/periodic_table.py --emulate-ranks N%s): the slowest shard PREDICTS the N-GPU wall time -- shards never "
                   "interact, the only collective is a gather of 64 doubles per atom" % (" --sweeps tolerance --poisson tolerance" if mode == "tolerance" else ""),
           "mode": mode,
           "predicted_seconds": {str(r["emulated_ranks"]): r["predicted_n_gpu_seconds"] for r in runs},
           "sum_of_shard_seconds": {str(r["emulated_ranks"]): r["sum_of_shard_seconds"] for r in runs},
           "runs": [{"emulated_ranks": r["emulated_ranks"], "shards": [{k: s[k] for k in ("rank", "atoms", "steps", "seconds", "finished")} for s in r["shards"]]} for r in runs]}
    one = out["predicted_seconds"]["1"]
    out["speedup_vs_one_gpu"] = {k: one / v for k, v in out["predicted_seconds"].items()}
    with open(path_of(mode, tag), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path_of(mode, tag), out["predicted_seconds"])


def shards_of(mode):
    d = json.load(open(path_of(mode)))
    return [s for r in d["runs"] for s in r["shards"] if s["atoms"]]


def fit(mode):
    from dftatom_amd import sweep
    sh = shards_of(mode)
    A = np.array([sweep.shard_features(s["atoms"]) for s in sh], dtype=float)
    y = np.array([s["seconds"] * 1e3 for s in sh])
    # relative least squares: every shard counts alike
    w = 1.0 / y
    coef, *_ = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)
    pred = A @ coef
    rel = (pred - y) / y
    return coef, rel


if __name__ == "__main__":
    if "--merge" in sys.argv:
        i = sys.argv.index("--merge")
        mode = sys.argv[sys.argv.index("--mode") + 1]
        files = [a for a in sys.argv[i + 1:] if a.endswith(".json")]
        merge(files, mode, sys.argv[sys.argv.index("--tag") + 1] if "--tag" in sys.argv else None)
        sys.exit(0)
    for mode in ("exact", "tolerance"):
        if os.path.exists(path_of(mode)):
            coef, rel = fit(mode)
            print('%-9s (FLOOR_SMALL_MS, FLOOR_BIG_MS, JOB_MS) = (%.1f, %.1f, %.3f)   residuals: max %.1f %%, rms %.1f %% over %d shards'
                  % (mode, coef[0], coef[1], coef[2], 100 * np.max(np.abs(rel)), 100 * np.sqrt(np.mean(rel ** 2)), len(rel)))
