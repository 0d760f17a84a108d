import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "diffab-pytorch_amd"), os.path.join(REPO, "oracle"), REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))

    return load


def maxrel(a, b):
    """max |a-b| / max |b|  (the metric used for every fp32 parity statement in this repo)."""
    import torch

    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def elemrel(a, b, floor=1e-2):
    """Element-wise relative error max |a-b|/|b| over the elements with |b| > floor * max|b| (north_star words its 1e-4 bar
    per value; `maxrel` above is the tensor-global norm, this one checks that no significant element hides behind it).
    The floor (1 % of the tensor maximum) is where an element-wise 1e-4 is still above the reference's OWN fp32 noise:
    BASELINE.md section 2 measures 1e-7..1e-6 of the tensor maximum between its fp32 and fp64 runs (and between 1 and 8 threads),
    i.e. up to 1e-4 relative on an element of 1 % of the maximum - below that an element-wise 1e-4 asks for more than the reference
    delivers.  `elemrel_by_decade` reports what happens below the floor."""
    import torch

    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    big = b.abs() > floor * b.abs().max()
    return float(((a - b).abs()[big] / b.abs()[big]).max())


def elemrel_by_decade(a, b, decades=5):
    """{k: worst |a-b|/|b| over the elements with 10^-(k+1) < |b| / max|b| <= 10^-k}, k = 0 .. decades-1: the element-wise error
    per decade of magnitude, so the claim behind `elemrel`'s floor is visible in the test log."""
    import torch

    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    r = b.abs() / b.abs().max()
    out = {}
    for k in range(decades):
        sel = (r <= 10.0 ** -k) & (r > 10.0 ** -(k + 1))
        if sel.any():
            out[k] = float(((a - b).abs()[sel] / b.abs()[sel]).max())
    return out
