"""`from src.models.detect_any_sound.detect_any_sound import DASM` (recipes/audioset_strong/detect_any_sound/passt/main.py:18)."""
from transformer4sed_amd.dasm import DASM  # noqa: F401
