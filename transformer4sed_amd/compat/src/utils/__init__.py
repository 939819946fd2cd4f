"""`from src.utils import update_ema, Logger, DataParallelWrapper, ExponentialDown, count_parameters,
load_yaml_with_relative_ref` (recipes/desed/finetune/train.py:19, recipes/desed/mlm/mlm_passt/main.py:25)."""
import importlib.util
import os
import sys

import yaml

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
    if name in ("Logger", "BestModels"):
        for p in sys.path:
            cand = os.path.join(p, "src", "utils", "log.py")
            if os.path.exists(cand) and os.path.dirname(os.path.abspath(cand)) != os.path.dirname(os.path.abspath(__file__)):
                spec = importlib.util.spec_from_file_location("_ref_src_utils_log", cand)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                return getattr(mod, name)
    raise AttributeError(name)
