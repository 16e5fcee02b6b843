#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline numbers (run on the MI355X box, e.g.
#   gpurun --timeout 2400 -- 'bash profiles/collect.sh r06'
# ) for the headline workload AND the throughput workloads the README quotes.  Per workload four separate passes of the SAME
# command: kernel trace + stats, the two HBM-side PMC counters (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never
# combined with sys/runtime traces), and the SQ counters.  The profiled program comes directly after `--` (python3 bench.py ...).
# Raw output goes to gpurun_out/<tag>_<workload>/ (scratch); profiles/summarize.py turns it into the committed summaries
#   profiles/<tag>_<workload>_{bench_kernel_stats.csv, bench_under_rocprof.json, hbm_traffic.json, sq_counters.json}.
set -u
TAG=${1:-r06}
ONLY=${2:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run_one() {
    local name=$1; shift
    local ARGS="$*"
    if [ -n "$ONLY" ] && [ "$ONLY" != "$name" ]; then return; fi
    local OUT=gpurun_out/${TAG}_${name}
    mkdir -p "$OUT"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 bench.py $ARGS > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o bench -- python3 bench.py $ARGS > /dev/null 2> "$OUT/fetch.err"
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o bench -- python3 bench.py $ARGS > /dev/null 2> "$OUT/write.err"
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/sq" -o bench -- python3 bench.py $ARGS > /dev/null 2> "$OUT/sq.err"
    python3 profiles/summarize.py "$OUT" "${TAG}_${name}" "$ARGS"
    tail -2 "$OUT/trace.err"
}
# the headline workload, Rn LSDA and the tolerance mode on bench.py's OWN window (--steps 20 --warmup 5, what the driver's line is
# measured on): summarize.py averages the timed launches alone, bench.py prints that average next to its HIP-event one
run_one default      --steps 20 --warmup 5 --no-cpu --no-extras
run_one scan         --steps 5 --warmup 2 --no-cpu --no-extras --scan-sweeps
run_one scan_tol     --steps 5 --warmup 2 --no-cpu --no-extras --scan-sweeps --tolerance
run_one tolerance    --steps 20 --warmup 5 --no-cpu --no-extras --tolerance
run_one rn_lsda      --lsda --steps 20 --warmup 5 --no-cpu --no-extras
run_one scan_adaptive --steps 5 --warmup 2 --no-cpu --no-extras --scan-sweeps --adaptive
run_one batch12      --atoms 12 --steps 5 --warmup 2 --no-cpu --no-extras
run_one batch256     --atoms 256 --steps 3 --warmup 1 --no-cpu --no-extras
run_one batch256_scan     --atoms 256 --steps 3 --warmup 1 --no-cpu --no-extras --scan-sweeps
run_one batch256_scan_tol --atoms 256 --steps 3 --warmup 1 --no-cpu --no-extras --scan-sweeps --tolerance
run_one l20          --levels 20 --lsda --steps 2 --warmup 1 --no-cpu --no-extras
run_one l20_scan_tol --levels 20 --lsda --steps 2 --warmup 1 --no-cpu --no-extras --scan-sweeps --tolerance
run_one l20_batch16  --levels 20 --atoms 16 --lsda --steps 2 --warmup 1 --no-cpu --no-extras
run_one l20_batch16_scan_tol --levels 20 --atoms 16 --lsda --steps 2 --warmup 1 --no-cpu --no-extras --scan-sweeps --tolerance
ls profiles | grep "^${TAG}_" | head -40
# the summaries travel back through gpurun_out/ (profiles/ on the GPU box is scratch); the raw traces stay behind
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_* gpurun_out/profiles_out/ 2>/dev/null
for d in gpurun_out/${TAG}_*; do rm -rf "$d/trace" "$d/fetch" "$d/write" "$d/sq"; done
