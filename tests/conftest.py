import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden(dict):
    """npz fixture -> dict of torch tensors (0-d arrays become python scalars)."""

    def __init__(self, name):
        super().__init__()
        js = os.path.join(GOLDEN, name + ".json")
        if os.path.exists(js):
            import json
            self.update(json.load(open(js)))
            return
        with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
            for k in f.files:
                a = f[k]
                self[k] = torch.from_numpy(a.copy()) if a.ndim else a.item()


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return load


def max_abs(a, b):
    return (a.double() - b.double()).abs().max().item()


def gate(name, value, limit):
    """assert value <= limit AND leave the measured value on record (a gate that prints nothing cannot be told from a slack
    one): one line per gate on stdout (`pytest -s` / the failure report) and in gpurun_out/measured_gates.log when that directory exists."""
    line = "GATE %-72s measured %.3e  limit %.1e  (%.0f %% of the limit)" % (name, value, limit, 100.0 * value / limit if limit else 0.0)
    print(line)
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        try:
            with open(os.path.join(out, "measured_gates.log"), "a") as f:
                f.write(line + "\n")
        except OSError:
            pass
    assert value <= limit, line
