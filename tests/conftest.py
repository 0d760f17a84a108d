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


def elemrel(a, b, floor=0.05):
    """Element-wise relative error max |a-b|/|b| over the elements with |b| > floor * max|b| (north_star words its 1e-4 bar
    per value; `maxrel` above is the tensor-global norm, this one checks that no significant element hides behind it).
    The floor is where an element-wise 1e-4 is still above the reference's OWN fp32 noise: BASELINE.md section 2 measures
    1e-7..1e-6 of the tensor maximum between its fp32 and fp64 runs (and between 1 and 8 threads), i.e. up to 1e-3 relative on an
    element of 1e-3 max; at 5 % of the maximum that noise is 2e-5 of the element."""
    import torch

    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    big = b.abs() > floor * b.abs().max()
    return float(((a - b).abs()[big] / b.abs()[big]).max())
