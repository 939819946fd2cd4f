"""`from src.models.cnn_transformer.passt_cnn import PaSST_CNN` (recipes/desed/pmam/main.py:97, finetune/cnn_trans/setting.py:6)."""
from transformer4sed_amd.passt_cnn import PaSST_CNN  # noqa: F401
