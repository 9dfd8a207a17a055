import glob
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def golden_files(prefix=""):
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def load_golden(path):
    z = np.load(path, allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    d["name"] = os.path.basename(path)[:-4]
    return d


def trace_names(prefix=""):
    return [os.path.basename(p)[:-4] for p in golden_files(prefix) if not os.path.basename(p).startswith("feat_")]


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as om

    om.build()
    return om
