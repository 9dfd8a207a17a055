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
    return [os.path.basename(p)[:-4] for p in golden_files(prefix) if not os.path.basename(p).startswith(("feat_", "crc_", "model_", "replay_", "collect_"))]


def crc_names():
    return sorted(os.path.basename(p)[:-4] for p in golden_files("crc_"))


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle as om

    om.build()
    return om


def step_record_bytes(*, actions, pos, alive, jobdone, rewards, done, trunc, metrics, used=None, counts=None, timer_left=None) -> bytes:
    """Canonical byte image of one env-step (what the CRC fixtures hash): little-endian, fixed order."""
    parts = [
        np.asarray(actions, dtype=np.int8).tobytes(),
        np.asarray(pos, dtype=np.int8).tobytes(),
        np.asarray(alive, dtype=np.uint8).tobytes(),
        np.asarray(jobdone, dtype=np.uint8).tobytes(),
        np.asarray(rewards, dtype="<f8").tobytes(),
        bytes([int(bool(done)), int(bool(trunc))]),
        np.asarray(metrics, dtype="<i4").tobytes(),
    ]
    if used is not None:
        parts += [np.asarray(used, dtype=np.uint8).tobytes(), np.asarray(counts, dtype=np.uint8).tobytes(),
                  np.asarray([timer_left], dtype="<i2").tobytes()]
    return b"".join(parts)
