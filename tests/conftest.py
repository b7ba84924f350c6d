"""Test configuration.  `-m "not gpu"` runs on the CPU build container (oracle vs goldens, host
logic, ABI export check, gloo world_size-2); `-m gpu` runs on the MI355X box and calls the HIP
path through the C ABI.  torch is imported lazily inside the CPU tests only: the GPU tests must
not load torch's bundled HIP runtime next to libgcnx's."""
import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gcn-string_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def golden_batch(g):
    from gcnx import synth
    return synth.HostBatch(g["x"], g["rowptr"], g["colidx"], g.get("vals"), g["graph_ptr"], g["y"])


def rel_err(a, b):
    """max |a-b| / max |b|: the normalised error the 1e-4 fp32 bar of north_star is read in."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30)) if a.size else 0.0


@pytest.fixture(scope="session")
def ctx():
    import gcnx
    c = gcnx.Context(0)   # raises on a box without a GPU: there is no CPU fallback
    yield c
    c.close()
