#!/bin/bash
# same-box A/B of two library builds (DESIGN_LOG.md r6 item 3): bash profiles/ab_libs.sh libA.so libB.so [reps]
# (both under dftatom_amd/; the binding loads $DFTA_LIB_PATH; alternating runs of the headline workload, level phase / multigrid / rounds per step)
mkdir -p gpurun_out/r6ab
for rep in $(seq 1 ${3:-3}); do
for lib in $1 $2; do
  DFTA_LIB_PATH=$PWD/dftatom_amd/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-extras > "gpurun_out/r6ab/b.json" 2> "gpurun_out/r6ab/b.err"
  python - <<PY
import json
d=json.loads(open("gpurun_out/r6ab/b.json").read().strip().splitlines()[-1])
print("$lib", d["ms_per_step"], d["phase_ms_per_step"]["levels"], d["phase_ms_per_step"]["poisson"], d["rounds_per_step"], d["etotal_last_step"])
PY
done
done
