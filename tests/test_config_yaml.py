"""The configurations `bench.py` times and the trainer fixtures replay are RESTATEMENTS of the reference's YAMLs (no YAML travels to the GPU box):
every value restated here must be the value in the reference's file.  Runs where the reference checkout exists (the build container); the
YAMLs themselves are consumed unchanged by the drop-in classes."""
import os
import sys

import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "config")), reason="reference checkout not present")


def load(rel):
    with open(os.path.join(REF, rel)) as f:
        return yaml.safe_load(f)


def subset_mismatches(mine, ref, path=""):
    """every leaf of `mine` must exist in `ref` with the same value (numbers compared as floats: YAML reads 5.0e-6 as a float, 1e-4 as a string)"""
    bad = []
    if isinstance(mine, dict):
        if not isinstance(ref, dict):
            return [(path, "not a mapping in the YAML")]
        for k, v in mine.items():
            if k not in ref:
                bad.append((f"{path}.{k}", "missing in the YAML"))
            else:
                bad += subset_mismatches(v, ref[k], f"{path}.{k}")
        return bad
    if isinstance(mine, (list, tuple)):
        if not isinstance(ref, (list, tuple)) or len(mine) != len(ref):
            return [(path, f"{mine!r} vs {ref!r}")]
        for i, (a, b) in enumerate(zip(mine, ref)):
            bad += subset_mismatches(a, b, f"{path}[{i}]")
        return bad
    try:
        same = float(mine) == float(ref) if not isinstance(mine, (str, bool)) and not isinstance(ref, bool) else mine == ref
    except (TypeError, ValueError):
        same = mine == ref
    return [] if same else [(path, f"{mine!r} vs {ref!r}")]


def test_bench_configurations_restate_the_reference_yamls():
    import bench
    ft2, ft1, pre, pmam = (load(p) for p in ("config/mat-sed/base/finetune2.yaml", "config/mat-sed/base/finetune1.yaml",
                                              "config/mat-sed/base/pretrain.yaml", "config/pmam/post_pretrain.yaml"))
    assert subset_mismatches(bench.FINETUNE2, ft2) == []
    assert subset_mismatches(bench.FINETUNE1, ft1) == []
    assert subset_mismatches(bench.PMAM, pmam) == []
    # pretrain.yaml keeps its schedule flat under `training` / `opt`, its batch under `generals` and the model kwargs directly under PaSST_SED
    # (mlm_passt/main.py reads them there); bench.py restates the same values in the finetune layout its trainer reads
    P = bench.PRETRAIN
    assert subset_mismatches({k: v for k, v in P["training"].items() if k not in ("batch_size", "scheduler")}, pre["training"]) == []
    assert P["training"]["batch_size"] == pre["generals"]["batch_size"]
    sc = P["training"]["scheduler"]
    assert (sc["n_epochs"], sc["n_epochs_cut"], sc["lr_warmup_rate"], sc["lr_warmup_epochs"], sc["exponent"]) == \
           (pre["training"]["n_epochs"], pre["training"]["n_epochs_cut"], pre["training"]["lr_warmup_rate"], pre["training"]["lr_warmup_epochs"],
            pre["opt"]["exponent"])
    assert subset_mismatches(P["PaSST_SED"]["init_kwargs"], pre["PaSST_SED"]) == []
    assert subset_mismatches(P["opt"]["param_groups"], pre["opt"]["param_groups"]) == []


def test_trainer_fixture_configurations_restate_the_reference_yamls():
    """oracle/make_golden.py's fixture configurations: same values as the YAMLs except what a miniature fixture must change (batch sizes; the
    flat `scheduler` block lives in *_SCHED with fixture epoch lengths)."""
    from oracle import make_golden as mg
    ft2, ft1, pre, pmam = (load(p) for p in ("config/mat-sed/base/finetune2.yaml", "config/mat-sed/base/finetune1.yaml",
                                              "config/mat-sed/base/pretrain.yaml", "config/pmam/post_pretrain.yaml"))

    def without_batch(cfg):
        c = {k: (dict(v) if isinstance(v, dict) else v) for k, v in cfg.items()}
        c["training"] = {k: v for k, v in cfg["training"].items() if k != "batch_size"}
        return c
    assert subset_mismatches(without_batch(mg.TRAINSTEP_CFG), ft2) == []
    assert subset_mismatches(without_batch(mg.TRAINSTEP_FT1_CFG), ft1) == []
    assert subset_mismatches(without_batch(mg.MLMSTEP_CFG), pre) == []
    # the PMAM fixture: post_pretrain.yaml's values except three a depth-2 miniature changes on purpose (recorded in its config_json)
    assert sorted(p_ for p_, _ in subset_mismatches(mg.PMAMSTEP_CFG, pmam)) == sorted(
        [".training.batch_size[0]", ".training.batch_size[1]", ".training.batch_size[2]",      # 2 + 2 + 2 clips instead of 6 + 6 + 12
         ".opt.param_groups.passt.lr",                                                           # 5e-5 instead of 5e-6: three steps must move the LoRA probes visibly
         ".opt.param_groups.passt.freeze_layer"])                                                # 1 of 2 blocks instead of 8 of 12
    pft2 = load("config/pmam/finetune2.yaml")
    assert subset_mismatches(without_batch(mg.PMAMFTSTEP_CFG), pft2) == []
    ys = pft2["training"]["scheduler"]
    sc = mg.PMAMFTSTEP_SCHED
    assert (sc["n_epochs"], sc["n_epochs_cut"], float(sc["exponent"]), sc["warmup_epochs"], float(sc["warmup_rate"])) == \
           (ys["n_epochs"], ys["n_epochs_cut"], float(ys["exponent"]), ys["lr_warmup_epochs"], float(ys["lr_warmup_rate"]))
    flat = dict(pre["training"], exponent=pre["opt"]["exponent"])          # (pretrain.yaml's flat layout)
    for sc, ys in ((mg.TRAINSTEP_SCHED, ft2["training"]["scheduler"]), (mg.TRAINSTEP_FT1_SCHED, ft1["training"]["scheduler"]), (mg.MLMSTEP_SCHED, flat)):
        assert (sc["n_epochs"], sc["n_epochs_cut"], float(sc["exponent"]), sc["warmup_epochs"], float(sc["warmup_rate"])) == \
               (ys["n_epochs"], ys["n_epochs_cut"], float(ys["exponent"]), ys["lr_warmup_epochs"], float(ys["lr_warmup_rate"])), (sc, ys)


@pytest.mark.parametrize("rel", ["mat-sed/base/finetune1.yaml", "mat-sed/base/finetune2.yaml", "mat-sed/base/pretrain.yaml",
                                 "pmam/post_pretrain.yaml", "pmam/finetune1.yaml", "pmam/finetune2.yaml"])
def test_drop_in_classes_take_the_reference_yamls_unchanged(rel):
    """`PaSST_SED(**configs["PaSST_SED"]["init_kwargs"])` / `PaSST_CNN(**configs["PaSST_CNN"]["init_kwargs"])` as the recipes' setting.py files
    call them (recipes/desed/finetune/passt/setting.py:5-12, mlm_passt/passt_mlm_setting.py:5-8, pmam/main.py:97): every key of every shipped
    YAML is accepted (construction needs no GPU; the only addition is `load_pretrained_model=False`, there is no checkpoint here), and the
    parameter count is the reference model's."""
    from transformer4sed_amd.passt_cnn import PaSST_CNN
    from transformer4sed_amd.passt_sed import PaSST_SED
    y = load("config/" + rel)
    if "PaSST_CNN" in y:
        kw = dict(y["PaSST_CNN"].get("init_kwargs", y["PaSST_CNN"]))
        kw["passt_sed_param"] = dict(kw["passt_sed_param"], load_pretrained_model=False)
        net = PaSST_CNN(**kw)
        want = {"pmam/post_pretrain.yaml": 97847387}.get(rel, 96180035)
    else:
        kw = dict(y["PaSST_SED"].get("init_kwargs", y["PaSST_SED"]), load_pretrained_model=False)
        net = PaSST_SED(**kw)
        want = 102129714 if kw.get("mlm") else 100947762          # 100.95 M (the finetune model), + the MLM head and mask token
    assert sum(p.numel() for p in net.parameters()) == want
    assert net.get_model_name() == ("PaSST_CNN" if "PaSST_CNN" in y else "PaSST_SED")
