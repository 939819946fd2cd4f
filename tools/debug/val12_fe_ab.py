"""Debug aid: the val12 parity test with the round-5 frontend kernel (do_log bit 1) vs the round-6 one."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, pytest
from transformer4sed_amd import frontend, ops
orig = ops.call
which = sys.argv[1]
def call(name, *a):
    if name == "sed_logmel_fwd" and which == "r5":
        a = a[:-1] + (a[-1] | 2,)
    return orig(name, *a)
frontend.call = call
import test_gpu_model as T
g = lambda name: np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"), allow_pickle=False)
try:
    T.test_validation_config_depth12_vs_reference_golden(g, "exact", None)
    print(which, "PASS")
except AssertionError as e:
    print(which, "FAIL", str(e)[:100])
