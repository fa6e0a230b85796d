"""Records how far inside its bound every tolerance check of the GPU parity tests lands (``within``), so that the bounds can
be kept at measured-plus-margin instead of by habit.  ``tests/conftest.py`` writes the session's records to
``gpurun_out/parity_margins.json`` when the run ends (the GPU box merges that directory back)."""
import json
import os

_RECORDS = []


def within(value, bound, label=""):
    """assert value < bound, remembering (label, value, bound)"""
    v = float(value)
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0]
    _RECORDS.append({"test": test, "label": str(label), "value": v, "bound": float(bound)})
    assert v < bound, (label, v, bound)
    return v


def dump(root):
    if not _RECORDS:
        return
    out = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_margins.json"), "w") as f:
            json.dump(_RECORDS, f, indent=0)
    except OSError:
        pass
