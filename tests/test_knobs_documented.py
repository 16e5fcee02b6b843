"""CPU suite: the debug / measurement knobs of the library (`dfta_knob("NAME")` in dftatom_amd/csrc and dftatom_amd/compat) and the list in
DESIGN.md section 9 are the same set -- a knob nobody can find is a hidden mode, a documented knob that no longer exists is a lie.
(Knobs never change results; the GPU tests prove that for each of them.)"""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_knob_is_documented_and_every_documented_knob_exists():
    in_code = set()
    for pat in ("dftatom_amd/csrc/*", "dftatom_amd/compat/*"):
        for f in glob.glob(os.path.join(ROOT, pat)):
            if f.endswith((".hip", ".h", ".inc", ".cpp")):
                in_code |= set(re.findall(r'dfta_knob\("([A-Z0-9_]+)"\)', open(f).read()))
    in_code.discard("NAME")                       # the example in common.h's comment
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    sec = design[design.index("## 9. Debug / measurement knobs"):]
    in_doc = {n for n in re.findall(r"`([A-Z][A-Z0-9_]+)(?:=[^`]*)?`", sec) if not n.startswith("DFTA_")}
    assert in_code - in_doc == set(), "knobs missing from DESIGN.md section 9: %s" % sorted(in_code - in_doc)
    assert in_doc - in_code == set(), "DESIGN.md section 9 names knobs the sources do not read: %s" % sorted(in_doc - in_code)
    assert len(in_code) >= 50
