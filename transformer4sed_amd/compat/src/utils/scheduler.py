"""`src.utils.scheduler`: ExponentialDown / update_ema come from this package; every other name of the reference module
(ExponentialWarmup, CosineDown, ...) is resolved lazily from the checkout's own file (src/utils/scheduler.py:8-39,79-122)."""
import importlib.util
import os
import sys

from transformer4sed_amd.scheduler import ExponentialDown, update_ema  # noqa: F401

_ref = None


def __getattr__(name):
    global _ref
    if name.startswith("__"):
        raise AttributeError(name)
    if _ref is None:
        here = os.path.abspath(__file__)
        for p in sys.path:
            cand = os.path.join(p, "src", "utils", "scheduler.py")
            if os.path.exists(cand) and os.path.abspath(cand) != here:
                spec = importlib.util.spec_from_file_location("_reference_src_utils_scheduler", cand)
                _ref = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(_ref)
                break
        else:
            raise AttributeError(name)
    return getattr(_ref, name)
