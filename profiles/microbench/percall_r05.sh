#!/bin/bash
# The reference's UNMODIFIED L3 (DFTAtom.cpp, linked against dftatom_amd/compat by `make -C oracle ref_l3` in the build container; the
# binary travels to the GPU box) on the per-call surface of DFT::Numerov: with the call-stream speculation of round 5 (compat/call_stream.h)
# and with one trial per call ($DFTA_COMPAT_NOSPECULATE) -- same printed protocol, time per SCF step.
#     gpurun -- 'bash profiles/microbench/percall_r05.sh > gpurun_out/percall_r05.txt'
EXE=oracle/_ref/ref_l3_cli
mkdir -p gpurun_out
$EXE 18 12 0.5 25 0.002 0 > gpurun_out/l3_ar12_spec.txt 2>&1
DFTA_COMPAT_NOSPECULATE=1 $EXE 18 12 0.5 25 0.002 0 > gpurun_out/l3_ar12_nospec.txt 2>&1
cmp gpurun_out/l3_ar12_spec.txt gpurun_out/l3_ar12_nospec.txt && echo "Ar @ 12 levels, $(wc -l < gpurun_out/l3_ar12_spec.txt) lines: speculation on / off IDENTICAL"
dftatom_amd/compat/dftatom_cli 18 12 0.5 25 0.002 0 chained > gpurun_out/cli_ar12_chained.txt 2>&1
cmp gpurun_out/l3_ar12_spec.txt gpurun_out/cli_ar12_chained.txt && echo "... and IDENTICAL to dftatom_cli in chained mode (the device-resident orchestrator)"
python3 - <<'PY'
import subprocess, time, os
exe = "oracle/_ref/ref_l3_cli"
runs = {}
for env in ({}, {"DFTA_COMPAT_NOSPECULATE": "1"}, {"DFTA_COMPAT_SWEEPS": "tolerance"}, {"DFTA_COMPAT_SWEEPS": "tolerance", "DFTA_COMPAT_NOSPECULATE": "1"}):
    slow = "DFTA_COMPAT_NOSPECULATE" in env and "DFTA_COMPAT_SWEEPS" not in env
    p = subprocess.Popen([exe, "86", "17", "0.5", "50", "0.0001", "0"], stdout=subprocess.PIPE, text=True, env=dict(os.environ, **env))
    stamps, lines = [], []
    for ln in p.stdout:
        lines.append(ln)
        if ln.startswith("Step:"):
            stamps.append(time.time())
            if len(stamps) >= (4 if slow else 14): break
    p.kill()
    tag = ", ".join("%s=%s" % kv for kv in sorted(env.items())) or "exact kernels, call-stream speculation (default)"
    runs[tag] = lines
    if slow:
        print("ref_l3_cli Rn @ 131 073 nodes [%s]: %.2f s per SCF step (%d steps timed)" % (tag, (stamps[-1] - stamps[1]) / (len(stamps) - 2), len(stamps) - 2))
    else:       # the history of the level end points (spines) needs three earlier searches of a level: steps 2 - 4 run without it
        print("ref_l3_cli Rn @ 131 073 nodes [%s]: %.2f s per SCF step in steps 2 - 4, %.2f s in steps 6 - 13" % (tag, (stamps[4] - stamps[1]) / 3, (stamps[13] - stamps[5]) / 8))
a, b = runs["exact kernels, call-stream speculation (default)"], runs["DFTA_COMPAT_NOSPECULATE=1"]
n = min(len(a), len(b))
print("Rn, exact kernels: the first %d printed lines with and without speculation identical: %s" % (n, a[:n] == b[:n]))
PY
for a in "18 12 0.002 25 5" "86 14 0.0005 25 15" "86 17 0.0001 50 15"; do echo "percall_levels $a: $(dftatom_amd/compat/percall_levels $a | tail -1)"; done
# LSDA and the uniform grid (modes 1, 2, 3 of ref_l3_cli): whole runs, printed text with / without the speculation
for args in "6 12 0.5 25 0.002 1" "10 13 0.5 12 0 2" "3 12 0.5 12 0 3" "29 12 0.5 25 0.002 1"; do
  s=$(date +%s%N); timeout 600 $EXE $args > gpurun_out/l3_a.txt 2>&1; m=$(date +%s%N)
  DFTA_COMPAT_NOSPECULATE=1 timeout 900 $EXE $args > gpurun_out/l3_b.txt 2>&1; e=$(date +%s%N)
  if cmp -s gpurun_out/l3_a.txt gpurun_out/l3_b.txt; then r=IDENTICAL; else r=DIFFERENT; fi
  echo "ref_l3_cli $args: $(wc -l < gpurun_out/l3_a.txt) lines $r; $(( (m - s) / 1000000 )) ms with speculation, $(( (e - m) / 1000000 )) ms without"
done
