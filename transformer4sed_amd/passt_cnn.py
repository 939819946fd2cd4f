"""`PaSST_CNN` -- drop-in for the reference's PMAM model class (src/models/cnn_transformer/passt_cnn.py:9-91): PaSST encoder with
LoRA linears, a 10-layer CNN branch, attention frequency pooling and a 384-wide Transformer-XL context network.  Constructor
kwargs (`passt_sed_param`, `cnn_param`), forward signature / return values, parameter and buffer names (state_dict interchange,
including the BatchNorm running statistics and the eval-mode LoRA weight folding) follow the reference; forward runs on the HIP
kernels of pmam_engine.py.  The nn.Modules are parameter containers only."""
import re

import torch
import torch.nn as nn

from .passt_sed import PaSST_SED, _Holder
from .pmam_engine import PmamEngine


class _CG(_Holder):
    def __init__(self, n):
        super().__init__()
        self.linear = nn.Linear(n, n)


class _CNN(_Holder):
    """Parameter layout of `CNN` (src/models/cnn/base.py:33-98) for activation 'cg' / normalization 'batch'."""

    def __init__(self, n_in_channel, nb_filters):
        super().__init__()
        self.cnn = nn.Sequential()
        cin = n_in_channel
        for i, co in enumerate(nb_filters):
            self.cnn.add_module(f"conv{i}", nn.Conv2d(cin, co, 3, 1, 1))
            self.cnn.add_module(f"batchnorm{i}", nn.BatchNorm2d(co, eps=0.001, momentum=0.99))
            self.cnn.add_module(f"cg{i}", _CG(co))
            cin = co


class PaSST_CNN(PaSST_SED):
    def __init__(self, passt_sed_param, cnn_param):
        super().__init__(**passt_sed_param, _pmam=True)
        if cnn_param is None:
            raise NotImplementedError("PaSST_CNN without the CNN branch is PaSST_SED")
        cp = dict(cnn_param)
        bad = []
        if cp.pop("cnn_name", "base") != "base": bad.append("cnn_name != 'base'")
        if cp.get("activation", "Relu").lower() != "cg": bad.append("activation != 'cg'")
        if cp.get("normalization", "batch") != "batch": bad.append("normalization != 'batch'")
        n = len(cp["nb_filters"])
        if cp.get("n_in_channel", 1) != 1: bad.append("n_in_channel != 1")
        if list(cp["kernel_size"]) != [3] * n or list(cp["padding"]) != [1] * n or list(cp["stride"]) != [1] * n:
            bad.append("only 3x3 / pad 1 / stride 1 convolutions")
        if any(c % 16 for c in cp["nb_filters"]): bad.append("filter counts must be multiples of 16")
        if "cnn_1d_dict" in cp: bad.append("cnn_1d_dict")
        if bad:
            raise NotImplementedError("the HIP PaSST_CNN path covers the PMAM configs only; unsupported: " + ", ".join(bad))
        self.cnn_filters = tuple(cp["nb_filters"])
        self.cnn_pooling = tuple(tuple(p) for p in cp["pooling"])
        self.conv_dropout = float(cp.get("conv_dropout", 0) or 0)
        fr = 128
        for _, pw in self.cnn_pooling:
            fr //= pw
        if fr != 1:
            raise NotImplementedError("the CNN branch must pool the 128 mel bins down to 1 (passt_cnn.py:53)")
        self.cnn = _CNN(1, self.cnn_filters)
        self.cnn_feat_dim = self.cnn_filters[-1]
        self.cnn_projector = nn.Linear(self.cnn_feat_dim, self.decoder_dim)
        self.merge_weight = nn.Parameter(torch.tensor([0.5]), requires_grad=bool(self.mlm))
        self.transformer_projector = nn.Linear(self.embed_dim, self.decoder_dim)
        self._drop_masks = None      # tests may inject the per-layer dropout masks ([B*H*W, C] uint8, layer order)
        self._mask_always_effective = True   # the merged sequence is contiguous: mask.py:66-80 writes in place (cf. DESIGN quirk 15)
        self._index_params()

    def _make_engine(self):
        return PmamEngine(self)

    def _grad_names(self):
        """Parameters the PMAM losses reach (the reference leaves `.grad` None on the others): the classifier is unused in MLM mode;
        the AT head, the final norm and the encoder blocks above the feature layer only through `at_out`."""
        at = getattr(self, "_at_grad_seen", False)
        names = set()
        for n, p in self._param_by_name.items():
            if not p.requires_grad or n.startswith("backbone.head") or (self.mlm and n.startswith("classifier.")):
                continue
            if n.startswith(("at_adpater", "backbone.norm.")) and not at:
                continue
            mt = re.match(r"backbone\.blocks\.(\d+)\.", n)
            if mt and int(mt.group(1)) >= self.passt_feature_layer and not at:
                continue
            names.add(n)
        return names

    def get_model_name(self):
        return "PaSST_CNN"
