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


def elem_err(a, b):
    """The smallest t for which np.allclose(a, b, rtol=t, atol=t * max|b|) holds: max_i |a_i - b_i| / (|b_i| + max|b|).
    The element-wise reading of "1e-4 relative" next to rel_err's max-norm reading: never larger than
    rel_err, and at least half of it."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if not a.size:
        return 0.0
    top = max(float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a - b) / (np.abs(b) + top)))


def strict_rel_err(a, b, floor=1e-3):
    """max_i |a_i - b_i| / |b_i| over the entries with |b_i| >= floor * max|b| (informational: the purely element-wise
    relative error where it is meaningful in fp32; entries below the floor are covered by rel_err / elem_err only)."""
    a, b = np.asarray(a, np.float64).ravel(), np.asarray(b, np.float64).ravel()
    if not a.size:
        return 0.0
    keep = np.abs(b) >= floor * max(float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a[keep] - b[keep]) / np.abs(b[keep]))) if keep.any() else 0.0


PARITY_REPORT = []     # (what, rel_err, elem_err, strict_rel_err, tol) of every assert_close call of the session


def assert_close(a, b, tol, what=""):
    """Both readings of the tolerance: max-norm (rel_err < tol) and element-wise (allclose(rtol=tol, atol=tol * max|ref|));
    the three error figures are kept for the session's parity report (gpurun_out/parity_report.json on the GPU box)."""
    a64, b64 = np.asarray(a, np.float64), np.asarray(b, np.float64)
    r, e, s = rel_err(a64, b64), elem_err(a64, b64), strict_rel_err(a64, b64)
    PARITY_REPORT.append({"what": str(what), "rel_err": r, "elem_err": e, "strict_rel_err_above_1e-3": s, "tol": tol})
    top = max(float(np.max(np.abs(b64))), 1e-30) if b64.size else 1.0
    assert r < tol, (what, "max-norm", r, tol)
    assert np.allclose(a64, b64, rtol=tol, atol=tol * top), (what, "element-wise", e, tol)


def pytest_sessionfinish(session, exitstatus):
    if not PARITY_REPORT:
        return
    try:
        import json
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_report.json"), "w") as fh:
            json.dump(PARITY_REPORT, fh, indent=1)
    except OSError:
        pass


@pytest.fixture(scope="session")
def ctx():
    import gcnx
    c = gcnx.Context(0)   # raises on a box without a GPU: there is no CPU fallback
    yield c
    c.close()
