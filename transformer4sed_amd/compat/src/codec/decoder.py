"""`src.codec.decoder` for the reference recipes: the two decode functions on the evaluation path run on the device
(transformer4sed_amd.evaluation); any other name of the reference module (e.g. the MAESTRO helpers) is resolved lazily from
the checkout's own file, which needs the third-party sed_scores_eval package only at that point."""
import importlib.util
import os
import sys

from transformer4sed_amd.evaluation import batched_decode_preds, decode_pred_batch_fast  # noqa: F401

_ref = None


def __getattr__(name):
    global _ref
    if name.startswith("__"):      # the import machinery probes __path__ / __spec__ ...: that must not load the reference module
        raise AttributeError(name)
    if _ref is None:
        here = os.path.abspath(__file__)
        for p in sys.path:
            cand = os.path.join(p, "src", "codec", "decoder.py")
            if os.path.exists(cand) and os.path.abspath(cand) != here:
                spec = importlib.util.spec_from_file_location("_reference_src_codec_decoder", cand)
                _ref = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(_ref)
                break
        else:
            raise AttributeError(name)
    return getattr(_ref, name)
