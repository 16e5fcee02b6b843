"""CPU suite: the level-parallel form of the oracle (bench.py's CPU baseline of SURVEY.md section 8d ii, oracle/libdfta_oracle_omp.so)
returns the bits of its serial form: un-chained clamped brackets, every level solved by its own call into its own density buffer,
buffers added in level order."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys, json, ctypes as C
sys.path.insert(0, %r)
import _oracle as O
o = O.oracle()
out = {}
for T in (1, 4):
    o.dfo_set_level_threads(T)
    res = []
    for lsda in (0, 1):
        s = o.dfo_scf_create(lsda, 18, 12, 0.5, 25.0, 2e-3, 3)
        e = O.Energies()
        for _ in range(3):
            o.dfo_scf_step(s, C.byref(e))
        res.append([e.Etotal.hex(), e.Ekinetic.hex(), e.Exc.hex()] + [s.contents.la[i].E.hex() for i in range(s.contents.nla)])
        o.dfo_scf_destroy(s)
    out[str(T)] = res
out["threads"] = o.dfo_get_level_threads()
print(json.dumps(out))
"""


def test_level_parallel_oracle_equals_serial():
    env = dict(os.environ, DFTA_ORACLE_OMP="1")
    r = subprocess.run([sys.executable, "-c", CODE % os.path.join(ROOT, "tests")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["threads"] == 4                      # built with -fopenmp
    assert d["1"] == d["4"]
