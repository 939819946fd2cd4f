"""The `compat/` import seam (SURVEY 8(b) "Import surface"): with `transformer4sed_amd/compat` in front of a checkout on sys.path,
every name the reference's loops import resolves -- hot-path names to this package's classes / functions, everything else to the
checkout's own modules -- and nothing of the checkout is executed before it is asked for.  The "checkout" here is a stub tree
with the reference's module layout (recipes/desed/finetune/train.py:15-19, finetune/passt/main.py:17, mlm/mlm_passt/main.py:25,
mlm/mlm_passt/train.py:6-7, pmam/main.py:25-26); runs in a fresh interpreter so that `src` is not cached."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = {
    "src/__init__.py": "",
    "src/models/__init__.py": "",
    "src/models/passt/__init__.py": "",
    "src/models/lora/__init__.py": "",
    "src/models/lora/layers.py": "class LoRALayer:\n    pass\n",
    "src/codec/__init__.py": "",
    "src/codec/encoder.py": "class Encoder:\n    pass\n",
    # the real file imports sed_scores_eval at module level: executing it early is what the seam must avoid
    "src/codec/decoder.py": "import os\nopen(os.environ['SED_STUB_MARK'], 'a').write('decoder\\n')\n\ndef decode_maestro():\n    return 'ref-maestro'\n",
    "src/utils/__init__.py": "raise RuntimeError('the checkout package src.utils must be shadowed by the adapter')\n",
    "src/utils/log.py": "class Logger:\n    origin = 'ref'\n\nclass BestModels:\n    origin = 'ref'\n",
    "src/utils/scheduler.py": "class ExponentialWarmup:\n    origin = 'ref'\n\nclass CosineDown:\n    origin = 'ref'\n\nclass ExponentialDown:\n    origin = 'ref'\n",
    "src/utils/statistics/__init__.py": "",
    "src/utils/statistics/model_statistic.py": "def count_parameters(m):\n    return 'ref-count'\n",
    "src/functional/__init__.py": "",
    "src/functional/loss.py": "class MSELoss:\n    origin = 'ref'\n",
    "src/evaluation_measures.py": "def compute_psds_from_scores():\n    return 'ref-psds'\n\ndef log_sedeval_metrics():\n    return 'ref-sedeval'\n",
    "src/preprocess/__init__.py": "",
    "src/preprocess/dataset.py": "class StronglyLabeledDataset:\n    origin = 'ref'\n",
    "src/postprocess/__init__.py": "",
}

CHILD = r"""
import json, os, sys
out = {}
mark = os.environ["SED_STUB_MARK"]
def marked():
    return os.path.exists(mark) and "decoder" in open(mark).read()
import transformer4sed_amd.passt_sed as P, transformer4sed_amd.data_aug as A, transformer4sed_amd.filter as F
import transformer4sed_amd.evaluation as E, transformer4sed_amd.scheduler as S, transformer4sed_amd.frontend as FE
from src.models.passt.passt_sed import PaSST_SED
from src.models.sed_model import SEDModel
from src.models.passt.passt_feature_extraction import PasstFeatureExtractor
from src.preprocess.data_aug import mixup, frame_shift, feature_transformation
from src.postprocess.filter import median_filter_torch
from src.codec.decoder import batched_decode_preds, decode_pred_batch_fast
out["hot"] = [PaSST_SED is P.PaSST_SED, SEDModel is P.SEDModel, PasstFeatureExtractor is FE.PasstFeatureExtractor, mixup is A.mixup,
              frame_shift is A.frame_shift, feature_transformation is A.feature_transformation, median_filter_torch is F.median_filter_torch,
              batched_decode_preds is E.batched_decode_preds, decode_pred_batch_fast is E.decode_pred_batch_fast]
out["decoder_executed_early"] = marked()
from src.utils import update_ema, Logger, DataParallelWrapper, ExponentialDown, count_parameters, load_yaml_with_relative_ref
out["utils"] = [update_ema is S.update_ema, ExponentialDown is S.ExponentialDown, Logger.origin == "ref", callable(count_parameters),
                callable(load_yaml_with_relative_ref), callable(DataParallelWrapper)]
from src.utils.log import BestModels
from src.utils.statistics.model_statistic import count_parameters as cp2
from src.utils.scheduler import ExponentialWarmup, CosineDown
from src.utils.scheduler import ExponentialDown as ED2
out["fallthrough"] = [BestModels.origin == "ref", cp2(None) == "ref-count", ExponentialWarmup.origin == "ref", CosineDown.origin == "ref",
                      ED2 is S.ExponentialDown]
from src.functional.loss import MSELoss
from src.evaluation_measures import compute_psds_from_scores, log_sedeval_metrics
from src.codec.encoder import Encoder
from src.preprocess.dataset import StronglyLabeledDataset
from src.models.lora.layers import LoRALayer
from src.models.lora import mark_only_lora_as_trainable
from src.models.cnn_transformer.passt_cnn import PaSST_CNN
from src.models.detect_any_sound.detect_any_sound import DASM
import transformer4sed_amd.passt_cnn as PC, transformer4sed_amd.pmam_trainer as PT, transformer4sed_amd.dasm as DM
out["checkout"] = [MSELoss.origin == "ref", compute_psds_from_scores() == "ref-psds", log_sedeval_metrics() == "ref-sedeval",
                   StronglyLabeledDataset.origin == "ref", Encoder.__module__ == "src.codec.encoder", LoRALayer.__module__ == "src.models.lora.layers",
                   mark_only_lora_as_trainable is PT.mark_only_lora_as_trainable, PaSST_CNN is PC.PaSST_CNN, DASM is DM.DASM]
out["decoder_still_lazy"] = not marked()
import src.codec.decoder as D
out["maestro"] = D.decode_maestro()          # any other name of the reference module: loaded from the checkout now
out["decoder_loaded_on_demand"] = marked()
# config/mat-sed/base/finetune2.yaml:62-75 restated (no checkpoint in this container)
kw = {"passt_feature_layer": 10, "f_pool": "mean_pool", "decode_ratio": 10, "at_adapter": True, "decoder": "transformerXL",
      "decoder_layer_num": 3, "decoder_pos_emd_len": 1000, "mlm": False, "load_pretrained_model": False}
net = PaSST_SED(**kw)
out["model"] = [net.get_model_name(), isinstance(net, SEDModel), sum(p.numel() for p in net.parameters())]
print("RESULT " + json.dumps(out))
"""


def test_reference_import_surface_resolves_through_compat(tmp_path):
    stub = tmp_path / "checkout"
    for rel, body in STUB.items():
        f = stub / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(body)
    mark = tmp_path / "mark.txt"
    env = dict(os.environ, SED_STUB_MARK=str(mark), CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "transformer4sed_amd", "compat"), str(stub), ROOT])
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(CHILD)], capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    assert all(out["hot"]), out["hot"]
    assert out["decoder_executed_early"] is False and out["decoder_still_lazy"] is True
    assert all(out["utils"]), out["utils"]
    assert all(out["fallthrough"]), out["fallthrough"]
    assert all(out["checkout"]), out["checkout"]
    assert out["maestro"] == "ref-maestro" and out["decoder_loaded_on_demand"] is True
    assert out["model"][0] == "PaSST_SED" and out["model"][1] is True and out["model"][2] == 100947762
