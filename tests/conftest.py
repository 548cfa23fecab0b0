import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # A gpu-marked test on a box without a GPU is an error in how the suite was
    # invoked (-m "not gpu" is the CPU invocation); skip rather than fail.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
        return {k: f[k] for k in f.files}


def golden_params(rec, prefix):
    """Ordered {fcN.weight/bias: tensor} from a golden record, e.g. prefix 'g0.'."""
    from collections import OrderedDict
    names = sorted((k[len(prefix):] for k in rec if k.startswith(prefix)),
                   key=lambda n: (int(n[2]), 0 if n.endswith("weight") else 1))
    return OrderedDict((n, torch.from_numpy(rec[prefix + n]).clone()) for n in names)


@pytest.fixture(scope="session")
def golden():
    return load_golden
