"""`from src.utils import update_ema, Logger, DataParallelWrapper, ExponentialDown, count_parameters,
load_yaml_with_relative_ref` (recipes/desed/finetune/train.py:19, recipes/desed/mlm/mlm_passt/main.py:25)."""
import importlib.util
import os
import sys

import yaml

# hot-path overrides first (scheduler.py here), then the reference's own src/utils (log.py, statistics/, ...): the recipes import
# `src.utils.log` (finetune/passt/main.py:17) and `src.utils.statistics.model_statistic` (pmam/main.py:26) as sub-modules
__path__ = [os.path.dirname(os.path.abspath(__file__))]
_rel = __name__.replace(".", os.sep)
for _p in sys.path:
    _cand = os.path.join(_p, _rel)
    if os.path.isdir(_cand) and os.path.abspath(_cand) != __path__[0]:
        __path__.append(os.path.abspath(_cand))

from transformer4sed_amd.scheduler import ExponentialDown, update_ema  # noqa: F401


class DataParallelWrapper:
    """Same contract as src/utils/__init__.py:11-21; with one process per GPU there is no nn.DataParallel to unwrap."""

    def __init__(self, model):
        self.model = model

    def __getattr__(self, name):
        return getattr(getattr(self.model, "module", self.model), name)

    def __call__(self, *args, **kwargs):
        return self.model(*args, **kwargs)


def load_yaml_with_relative_ref(yaml_path) -> dict:
    with open(yaml_path, "r") as f:
        main = yaml.safe_load(f)
    if isinstance(main, dict) and "include" in main:
        inc = main.pop("include")
        base = load_yaml_with_relative_ref(inc["base_path"])
        for key in inc["keys"]:
            main[key] = base[key]
    return main


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def __getattr__(name):
    """Logger / BestModels live in the reference's src/utils/log.py (not on the hot path): load them lazily from there."""
    if name.startswith("__"):
        raise AttributeError(name)
    if name in ("Logger", "BestModels"):
        for p in sys.path:
            cand = os.path.join(p, "src", "utils", "log.py")
            if os.path.exists(cand) and os.path.dirname(os.path.abspath(cand)) != os.path.dirname(os.path.abspath(__file__)):
                spec = importlib.util.spec_from_file_location("_ref_src_utils_log", cand)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                return getattr(mod, name)
    raise AttributeError(name)
