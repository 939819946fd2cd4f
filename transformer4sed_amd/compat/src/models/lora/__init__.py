"""`from src.models.lora import mark_only_lora_as_trainable` (recipes/desed/pmam/main.py:25)."""
from transformer4sed_amd.pmam_trainer import mark_only_lora_as_trainable  # noqa: F401
