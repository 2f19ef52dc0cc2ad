import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def split(buf, off):
    """concatenated array + offsets -> list of slices"""
    return [buf[int(off[i]):int(off[i + 1])] for i in range(len(off) - 1)]


class Golden:
    """Lazy access to the committed reference outputs under tests/golden/."""

    def __init__(self):
        self._npz = {}

    def npz(self, name):
        if name not in self._npz:
            self._npz[name] = np.load(os.path.join(GOLDEN, name + ".npz"))
        return self._npz[name]

    def reads(self):
        z = self.npz("toy_reads")
        single = [x.tobytes() for x in split(z["single"], z["single_off"])]
        p1 = [x.tobytes() for x in split(z["p1"], z["p1_off"])]
        p2 = [x.tobytes() for x in split(z["p2"], z["p2_off"])]
        return single, p1, p2

    def db_path(self, name):
        return os.path.join(GOLDEN, name)

    def expected(self, db, key):
        z = self.npz(db + "_expected")
        return split(z[key], z[key + "_off"])


@pytest.fixture(scope="session")
def golden():
    return Golden()
