"""`from src.models.detect_any_sound.at_adapter import QueryBasedAudioTaggingDecoder` (detect_any_sound.py:14): on this path the query
decoder is part of `transformer4sed_amd.dasm.DasmHead` (csrc/dasm.hip)."""
from transformer4sed_amd.dasm import DasmHead as QueryBasedAudioTaggingDecoder  # noqa: F401
