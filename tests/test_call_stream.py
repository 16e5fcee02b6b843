"""CPU suite: dftatom_amd/compat/call_stream.h -- the mirror of the reference's per-call level search -- against a synthetic sweep
(tests/cpp/call_stream_check.cpp: no device).  The speculation may only ever change HOW a call is served: every answer must be the
direct evaluation's, for the reference's protocol, for a caller with another energyErr and for a caller without any pattern; for the
reference's protocol the launches must be a small fraction of the calls (one launch carries the tree of the next 12 - 13 decisions)."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_call_stream_serves_the_reference_protocol_from_its_cache(tmp_path):
    exe = str(tmp_path / "call_stream_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-o", exe, os.path.join(HERE, "cpp", "call_stream_check.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr                       # non-zero: an answer differed from the direct evaluation
    m = re.search(r"reference protocol: calls (\d+) launches (\d+) hits (\d+) trials (\d+) wrong (\d+)", r.stdout)
    calls, launches, hits, trials, wrong = (int(x) for x in m.groups())
    assert wrong == 0 and launches + hits == calls
    assert launches <= 0.10 * calls, (calls, launches)                  # 6 levels x ~147 calls in ~12 launches each
    assert trials <= 8191 * launches
    eig = [float(x) for x in re.findall(r"level \d+ E (\S+)", r.stdout)]
    want = [-3204.75642, -535.87331, -512.1183, -130.2447, -118.90112, -99.3301]
    assert len(eig) == 6 and all(abs(a - b) < 2e-12 * max(1.0, abs(b)) + 2e-12 for a, b in zip(eig, want))
    # the same levels searched again with slightly moved eigenvalues (a new stream each time, as the reference builds a new Numerov per SCF
    # step): from the fourth search on the process-wide history of the end points gives every launch a spine of predicted decisions
    m = re.search(r"history: launches first search (\d+) last search (\d+) wrong (\d+)", r.stdout)
    first, last, w = (int(x) for x in m.groups())
    assert w == 0 and last <= 0.8 * first, (first, last)
    for tag in ("other energyErr", "arbitrary caller"):
        m = re.search(tag + r": calls (\d+) launches (\d+) hits (\d+) wrong (\d+)", r.stdout)
        c, la, h, w = (int(x) for x in m.groups())
        assert w == 0 and la + h == c
    m = re.search(r"arbitrary caller: calls (\d+) launches (\d+) hits (\d+) wrong (\d+) trials (\d+)", r.stdout)
    assert int(m.group(2)) == 400                                      # nothing to predict: one launch per call ...
    assert int(m.group(5)) <= 12 * 4095 + 400                          # ... and, after three useless trees, single trials (pauses of 16, 32, 64 ... calls)
