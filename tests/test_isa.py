"""Code-generation guards for the out-of-line device functions (no GPU needed: hipcc cross-compiles gfx950 assembly here).

A pointer that reaches a non-inlined device function has no address space; without the explicit re-derivation of
`csrc/common.h` (dfta_as_global / dfta_as_constant / dfta_uniform) the compiler emits FLAT loads, whose waits are waits for every
outstanding access -- the scan sweeps then run 1.5x slower (round 5: 256 atoms 346 -> 275 ms per step).  The arithmetic is untouched
by this; the tests only look at which memory instructions the hot functions contain.
"""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "dftatom_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-function", "-S", "--cuda-device-only"]


def _functions(asm):
    """name -> list of instruction lines, for every function symbol of an AMDGPU assembly listing"""
    out, name = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z[A-Za-z0-9_]+):", line)
        if m:
            name = m.group(1)
            out[name] = []
        elif line.startswith(".Lfunc_end"):
            name = None
        elif name and line.startswith("\t") and not line.startswith("\t."):
            out[name].append(line.strip())
    return out


def _count(lines, prefix):
    return sum(1 for ln in lines if ln.startswith(prefix))


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_scan_sweeps_read_their_tables_with_global_and_scalar_loads(tmp_path):
    out = tmp_path / "scan.s"
    subprocess.run([HIPCC] + FLAGS + ["-o", str(out), os.path.join(CSRC, "scan.hip")], check=True, capture_output=True, timeout=600)
    fns = _functions(out.read_text())
    sweeps = {n: ls for n, ls in fns.items() if "10scan_sweepIL" in n}
    assert len(sweeps) == 2, sorted(fns)          # CountNodes and SolutionInZero: real functions, called ~150 times per level
    for name, lines in sweeps.items():
        flat, glob, buf, scal = _count(lines, "flat_load"), _count(lines, "global_load"), _count(lines, "buffer_load"), _count(lines, "s_load")
        # the row loops: the veff rows come through buffer loads (scalar row offset + constant lane offset: no vector address arithmetic,
        # in-order vmcnt: 32 rows in flight per wave), the eight factors T of a batch through ONE scalar load; the other tables through
        # global loads; a handful of flat accesses to the caller's LaneState / ScanGrid copies in scratch are all that may remain
        assert buf > 300 and glob > 50 and scal > 30, (name, flat, glob, buf, scal)
        assert flat <= 16, (name, flat, glob, buf, scal)
        assert not any("v_readfirstlane" in ln and i + 12 < len(lines) and any("s_cbranch_execnz" in x for x in lines[i:i + 12]) and
                       any("buffer_load" in x for x in lines[i:i + 12]) for i, ln in enumerate(lines)), "waterfall loop around a buffer load: the descriptor is not wave-uniform"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_device_side_search_uses_global_accesses_out_of_line(tmp_path):
    out = tmp_path / "numerov.s"
    subprocess.run([HIPCC] + FLAGS + ["-o", str(out), os.path.join(CSRC, "numerov.hip")], check=True, capture_output=True, timeout=900)
    fns = _functions(out.read_text())

    def one(tag):
        hit = [ls for n, ls in fns.items() if tag in n]
        assert len(hit) == 1, (tag, [n for n in fns if tag in n])
        return hit[0]

    flat = lambda ls: _count(ls, "flat_load") + _count(ls, "flat_store") + _count(ls, "flat_atomic")
    cand, close, sweep = one("17persist_candidate"), one("13persist_close"), one("13persist_sweep")
    # the match solve inside both (helper wave loading ahead of the integrator) and the sweep pipeline's trial arrays: global
    assert _count(cand, "global_load") > 100 and flat(cand) <= 40, (_count(cand, "global_load"), flat(cand))
    assert _count(close, "global_load") > 150 and flat(close) <= 60, (_count(close, "global_load"), flat(close))
    assert _count(sweep, "global_load") > 40 and flat(sweep) <= 40, (_count(sweep, "global_load"), flat(sweep))
