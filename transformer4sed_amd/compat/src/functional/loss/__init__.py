"""`from src.functional.loss import loss_function_factory` (recipes/audioset_strong/base/passt_cnn/train.py:17, :47-50): the elementwise
supervised losses of the AudioSet-Strong recipes on the fused HIP kernel (transformer4sed_amd/dasm_trainer.py)."""
from transformer4sed_amd.dasm_trainer import loss_function_factory  # noqa: F401
