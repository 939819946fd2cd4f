import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The test processes also run the CPU oracle (torch fp32): give it every CPU the container may really use (affinity / cgroup quota)
# instead of the product's host-thread cap (transformer4sed_amd/hostcpu.py) or torch's default of half the visible cores.
os.environ.setdefault("SED_HOST_THREADS", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # (what an entry point sets before HIP initialises: transformer4sed_amd.hostcpu.recommended_env)


def pytest_configure(config):
    import torch
    from transformer4sed_amd.hostcpu import usable_cpus
    if "OMP_NUM_THREADS" not in os.environ:
        torch.set_num_threads(min(torch.get_num_threads(), usable_cpus()))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load
