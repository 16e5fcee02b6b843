"""Debug / measurement knobs for the tests: ONE environment variable, DFTA_DEBUG="NAME[=VALUE],..." (dftatom_amd/csrc/common.h:dfta_knob).
The separate DFTA_<NAME> variables of rounds 1-2 are gone; `knobs` keeps the tests' spelling (DFTA_POISSON_GROUP="0") and writes the entry."""
import os


class knobs:
    """with knobs(DFTA_POISSON_GROUP="0", LEVELS_STATIC="1"): ... -- the entries are appended to DFTA_DEBUG for the duration of the block.
    The library reads a knob when the object it belongs to is created (solver, SCF) unless a test says otherwise."""

    def __init__(self, mapping=None, **kv):
        self.kv = dict(mapping or {})
        self.kv.update(kv)

    def __enter__(self):
        self.old = os.environ.get("DFTA_DEBUG")
        entries = [e for e in (self.old or "").split(",") if e]
        for k, v in self.kv.items():
            name = k[5:] if k.startswith("DFTA_") else k
            entries.append("%s=%s" % (name, v))
        if entries:
            os.environ["DFTA_DEBUG"] = ",".join(entries)
        return self

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("DFTA_DEBUG", None)
        else:
            os.environ["DFTA_DEBUG"] = self.old
