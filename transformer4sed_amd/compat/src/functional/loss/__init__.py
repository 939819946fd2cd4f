"""`from src.functional.loss import loss_function_factory` (recipes/audioset_strong/base/passt_cnn/train.py:17, :47-50): the elementwise
supervised losses of the AudioSet-Strong recipes on the fused HIP kernel (transformer4sed_amd/dasm_trainer.py).  Every other name of the
reference module (`MSELoss` of recipes/desed/mlm/mlm_passt/train.py:6, SupConLoss, InfoNCE ...) is resolved lazily from the checkout's
own `src/functional/loss/__init__.py` (or a flat `loss.py`), whose sub-modules stay importable through the extended package path."""
import importlib.util
import os
import sys

from transformer4sed_amd.dasm_trainer import loss_function_factory  # noqa: F401

__path__ = [os.path.dirname(os.path.abspath(__file__))]
_ref = None
_ref_file = None
for _p in sys.path:
    _base = os.path.join(_p, "src", "functional")
    _pkg, _flat = os.path.join(_base, "loss"), os.path.join(_base, "loss.py")
    if os.path.isdir(_pkg) and os.path.abspath(_pkg) != __path__[0]:
        __path__.append(os.path.abspath(_pkg))
        if _ref_file is None and os.path.exists(os.path.join(_pkg, "__init__.py")):
            _ref_file = os.path.join(_pkg, "__init__.py")
    elif os.path.isfile(_flat) and _ref_file is None:
        _ref_file = _flat


def __getattr__(name):
    global _ref
    if name.startswith("__") or _ref_file is None:
        raise AttributeError(name)
    if _ref is None:
        spec = importlib.util.spec_from_file_location("_reference_src_functional_loss", _ref_file)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        _ref = mod
    return getattr(_ref, name)
