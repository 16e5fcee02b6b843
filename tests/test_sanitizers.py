"""CPU suite: the CPU-side code under AddressSanitizer + UndefinedBehaviorSanitizer.

`make -C oracle sanitize` builds oracle/dfta_oracle.c, dftatom_amd/csrc/ctx_grid.cpp (the product's host code, compiled with g++
against the HIP headers) and oracle/sanitize_main.cpp with -fsanitize=address,undefined; the driver exercises sweeps, the level
driver, a multigrid solve, VWN, the quadrature rules, Aufbau / spin split for Z = 1..118 (oracle vs product) and two SCF steps of
argon in LDA and LSDA.  Pass = exit code 0 and nothing on the sanitizers' report stream.  (GPU sanitizers are not available on the
pool; the HIP kernels are covered by the parity suite.)
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_and_host_code_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "sanitize"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([os.path.join(ROOT, "oracle", "_build", "sanitize_main")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "sanitize_main: ok" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-4000:]
