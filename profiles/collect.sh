#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline numbers (run on the MI355X box, e.g.
#   gpurun --timeout 1500 -- 'bash profiles/collect.sh r01'
# ).  Three separate passes of the SAME command: kernel trace + stats, then the two HBM-side PMC counters
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never combined with sys/runtime traces).
# Raw output goes to gpurun_out/<tag>_*/ (scratch); profiles/summarize.py turns it into the committed summaries.
set -u
TAG=${1:-r01}
ARGS=${2:-"--steps 3 --warmup 1 --no-cpu --no-extras"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/${TAG}
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- python3 bench.py $ARGS > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o bench -- python3 bench.py $ARGS > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o bench -- python3 bench.py $ARGS > /dev/null 2> "$OUT/write.err"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/sq" -o bench -- python3 bench.py $ARGS > /dev/null 2> "$OUT/sq.err"
python3 profiles/summarize.py "$OUT" "$TAG" "$ARGS"
ls -la "$OUT" "$OUT"/*/ 2>/dev/null | head -40
