#!/usr/bin/env python
"""MAT-SED train-step benchmark (BASELINE.json metric: clips/sec, 10 s clips, full train step).

  python bench.py --gpus N --steps K --warmup W          (N > 1 is launched by torch.distributed.run, one rank per GPU)

Workload (config.workload): the finetune2 mean-teacher step of config/mat-sed/base/finetune2.yaml on DESED-shaped
synthetic clips (320 000 samples @ 32 kHz): log-mel frontend -> frame_shift / mixup / freq-warp + FilterAugment (two views)
-> student forward+backward (all 100.95 M parameters trainable) -> EMA teacher forward with 11 sliding windows (no grad)
-> six BCE/MSE losses -> fused AdamW + EMA.  Per-GPU batch 32 (strong+synth 11 / weak 11 / unlabeled 10): weak scaling.
Weights are deterministic synthetic (no network for the PaSST checkpoint).  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before HIP initialises: the entry point's job (transformer4sed_amd/__init__.py, hostcpu.recommended_env)

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# values of config/mat-sed/base/finetune2.yaml (reference), restated: lines 11-38 (training), 62-86 (PaSST_SED), 88-100 (opt)
FINETUNE2 = {
    "training": {
        "batch_size": [3, 1, 4, 4], "ema_factor": 0.999, "w_weak": 0.5, "w_cons_max": 40, "w_cons_min": 0,
        "w_weak_cons": 0.5, "w_AT": 2, "self_loss_warmup": 15, "cons_scheduler_name": "Sigmoid",
        "scheduler": {"n_epochs": 30, "n_epochs_cut": 15, "exponent": -1, "lr_warmup_rate": 0.1, "lr_warmup_epochs": 1},
        "transform": {"n_transform": 2, "choice": [1, 0, 0, 1], "filter_db_range": [-26, 26], "filter_bands": [2, 5],
                      "filter_minimum_bandwidth": 4, "filter_type": "step"},
    },
    "PaSST_SED": {
        "init_kwargs": {"passt_feature_layer": 10, "f_pool": "mean_pool", "decode_ratio": 10, "at_adapter": True,
                        "decoder": "transformerXL", "decoder_layer_num": 3, "decoder_pos_emd_len": 1000, "mlm": False},
        "train_stu_kwargs": {"encoder_win": False, "win_param": [512, 49], "mix_rate": 0.5, "temp_w": 1},
        "train_tch_kwargs": {"encoder_win": True, "win_param": [512, 49], "mix_rate": 0.5, "temp_w": 1},
    },
    "opt": {"param_groups": {"encoder": {"lr": 5.0e-6, "weight_decay": 1.0e-4, "freeze_layer": 0, "step_lr": 4},
                             "decoder": {"lr": 1.0e-4, "weight_decay": 1.0e-4},
                             "head": {"lr": 1.0e-4, "weight_decay": 1.0e-4}}},
}
# config/mat-sed/base/finetune1.yaml (encoder + context net frozen, heads train; teacher without windows) -- lines 11-36, 70-95, 128-142
FINETUNE1 = json.loads(json.dumps(FINETUNE2))
FINETUNE1["training"].update({"self_loss_warmup": 8, "cons_scheduler_name": "Linear", "w_cons_max": 2,
                              "scheduler": {"n_epochs": 15, "n_epochs_cut": 10, "exponent": -1, "lr_warmup_rate": 0.1,
                                            "lr_warmup_epochs": 0}})
FINETUNE1["PaSST_SED"]["train_tch_kwargs"] = {"encoder_win": False, "win_param": [512, 49], "mix_rate": 0.5, "temp_w": 1}
FINETUNE1["opt"] = {"param_groups": {"encoder": {"lr": 0, "weight_decay": 1.0e-4, "freeze_layer": 0, "step_lr": 4},
                                     "decoder": {"lr": 0, "weight_decay": 1.0e-4}, "head": {"lr": 2.0e-4, "weight_decay": 1.0e-4}}}
# config/mat-sed/base/pretrain.yaml (masked-frame reconstruction, encoder frozen) -- lines 8-26, 41-54, 88-101
PRETRAIN = {
    "training": {"encoder_win": False, "batch_size": [4, 4, 16],
                 "scheduler": {"n_epochs": 15, "n_epochs_cut": 10, "exponent": -0.5, "lr_warmup_rate": 0.1, "lr_warmup_epochs": 1},
                 "transform": {"n_transform": 1, "choice": [1, 0, 0, 1], "filter_db_range": [-26, 26], "filter_bands": [2, 5],
                               "filter_minimum_bandwidth": 4, "filter_type": "step"}},
    "PaSST_SED": {"init_kwargs": {"passt_feature_layer": 10, "f_pool": "mean_pool", "decode_ratio": 10, "at_adapter": True,
                                  "decoder": "transformerXL", "decoder_layer_num": 3, "decoder_pos_emd_len": 1000, "mlm": True,
                                  "mlm_dict": {"strategy": "block", "block_width": 10, "mask_rate": 0.75, "out_dim": 768}}},
    "opt": {"param_groups": {"encoder": {"lr": 0, "weight_decay": 1.0e-4, "freeze_layer": 0, "step_lr": 4},
                             "decoder": {"lr": 1.0e-4, "weight_decay": 1.0e-4}, "head": {"lr": 1.0e-4, "weight_decay": 1.0e-4}}},
}
# config/pmam/post_pretrain.yaml (PMAM post-pretraining: PaSST + LoRA r=8, CNN branch, attention pooling, 384-wide context net,
# prototype-similarity BCE on the masked frames) -- lines 8-37 (training), 47-89 (PaSST_CNN), 105-122 (opt)
PMAM = {
    "training": {"batch_size": [6, 6, 12], "w_AT": 0.1, "clip_grad": True,
                 "scheduler": {"n_epochs": 30, "n_epochs_cut": 10, "exponent": -1.5, "lr_warmup_rate": 0.1, "lr_warmup_epochs": 0},
                 "transform": {"n_transform": 1, "choice": [1, 0, 0, 1], "filter_db_range": [-26, 26], "filter_bands": [2, 5],
                               "filter_minimum_bandwidth": 4, "filter_type": "step"}},
    "PaSST_CNN": {
        "init_kwargs": {
            "passt_sed_param": {"passt_feature_layer": 10, "class_num": 30, "f_pool": "attention", "decode_ratio": 10, "at_adapter": True,
                                "decoder": "transformerXL", "decoder_layer_num": 3, "decoder_pos_emd_len": 1000, "decoder_dim": 384,
                                "mlm": True, "lora_config": {"r": 8, "lora_alpha": 1, "requires_grad_pretrain": False},
                                "mlm_dict": {"strategy": "block", "block_width": 10, "mask_rate": 0.8, "out_dim": 768,
                                             "mask_style": [0.9, 0.05, 0.05]}},
            "cnn_param": {"n_in_channel": 1, "activation": "cg", "conv_dropout": 0.5, "kernel_size": [3] * 10, "padding": [1] * 10,
                          "stride": [1] * 10, "nb_filters": [16, 16, 32, 32, 64, 64, 128, 128, 256, 384],
                          "pooling": [[2, 2], [1, 1], [2, 2], [1, 1], [1, 2], [1, 2], [1, 2], [1, 2], [1, 2], [1, 1]]}},
        "train_kwargs": {"encoder_win": False, "temp_w": 1}},
    "opt": {"param_groups": {"cnn": {"lr": 1.5e-4, "weight_decay": 1.0e-4},
                             "passt": {"lr": 5.0e-6, "weight_decay": 1, "freeze_layer": 8, "step_lr": 0},
                             "decoder": {"lr": 1.5e-4, "weight_decay": 1.0e-4}, "head": {"lr": 2.0e-4}}},
}
# DASM training (recipes/audioset_strong/detect_any_sound/passt/train.py:66-120).  The reference ships no YAML for this recipe (and its
# main.py imports a module that does not exist): values in the style of config/pmam -- BCE on both streams, w_AT 0.5, temperature 0.5, the
# FilterAugment / frequency-warp transform of the other recipes, PMAM's learning rates with the whole encoder trainable.
DASM_TRAIN = {
    "training": {"w_AT": 0.5, "clip_grad": True,
                 "scheduler": {"n_epochs": 30, "n_epochs_cut": 10, "exponent": -1.5, "lr_warmup_rate": 0.1, "lr_warmup_epochs": 0},
                 "transform": {"n_transform": 1, "choice": [1, 0, 0, 1], "filter_db_range": [-26, 26], "filter_bands": [2, 5],
                               "filter_minimum_bandwidth": 4, "filter_type": "step"}},
    "class_loss": {"loss_name": "BCELoss", "kwargs": None},
    "DASM": {"train_kwargs": {"encoder_win": False, "temp_w": 0.5}},
    "opt": {"param_groups": {"cnn": {"lr": 1.5e-4, "weight_decay": 1.0e-4}, "passt": {"lr": 5.0e-6, "weight_decay": 1.0e-4, "freeze_layer": 0, "step_lr": 0},
                             "decoder": {"lr": 1.5e-4, "weight_decay": 1.0e-4}, "head": {"lr": 2.0e-4}}},
}
MODE_CFG = {"finetune2": FINETUNE2, "val": FINETUNE2, "finetune1": FINETUNE1, "pretrain": PRETRAIN, "pmam": PMAM, "dasm": PMAM, "dasm_train": DASM_TRAIN}
MODE_GFLOP = {"finetune2": (2463.16, 21.22), "finetune1": (649.13, 14.15), "pretrain": (383.68, 14.15), "val": (2 * 2264.3, 2 * 7.07),
              # PMAM post-pretrain step: FlopCounterMode on the reference's own PaSST_CNN (depth 12, LoRA r = 8, freeze_layer 8, forward + loss +
              # backward) at B = 1, 2 -- oracle/make_golden.py gen_pmamflops.  (The reference never forms the full dW of a LoRA layer; this
              # build does, as an intermediate of dA / dB: those FLOPs are not credited.)
              "pmam": (433.06, 3.54),
              "dasm": (None, None),       # (no FlopCounter figure of the reference's DASM inference was recorded)
              # DASM train step: FlopCounterMode on the reference's own DASM (depth 12, everything trainable, the 407 AudioSet-Strong classes
              # as learned queries, forward + BCE losses + backward) at B = 1, 2 -- oracle/make_golden.py gen_dasmflops
              "dasm_train": (961.73, 16.07)}
GFLOP_PER_CLIP = 2463.16      # finetune2 step, algorithmic GEMM+conv FLOPs per clip of the REFERENCE's schedule (a-term)
GFLOP_PER_BATCH = 21.22       # batch-shared linear_pos GEMMs (b-term)
# What this build does not execute: the teacher's 11 sliding windows stop after the tapped block 10 (blocks 11-12 of a window feed
# nothing, engine._encoder_fwd).  Per block and window of N tokens: 14.156 MFLOP/token of GEMMs + 4 N^2 768 of attention;
# 10 windows of 602 tokens + one of 590 -> 2 x (10 x 9.635 + 9.421) = 211.5 GFLOP per clip.
# validation: student AND teacher run the 17 windows of val_kwargs, each stopping after block 10: 2 nets x (16 windows of 602 tokens + one
# of 590) x 2 blocks = 2 x 2 x (16 x 9.635 + 9.421) = 654.3 GFLOP per clip never executed
GFLOP_SKIPPED = {"finetune2": 211.5, "val": 654.3}
PEAK_BF16_TFLOPS = 2500.0     # dense 16-bit MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0         # HBM3E peak, MI355X_MICROARCH.md
MEASURED_HBM_GBS = 6300.0     # achievable streaming rate measured on the part (same guide)
GEMM_KERNELS = ["sed_gemm_nt", "sed_gemm_qkv", "sed_gemm_qkv_w2s", "sed_gemm_dw_tn", "sed_gemm_nt_gb", "sed_gemm_qkv_gb", "sed_gemm_nt_w2", "sed_gemm_qkv_w2", "sed_gemm_nt_w2f8", "sed_gemm_qkv_w2f8", "sed_gemm_nt_gb_e4m3", "sed_gemm_nt_lnp", "sed_gemm_nt_lnp8", "sed_gemm_nt_lnc", "sed_gemm_qkv_lnc", "sed_gemm_nt_lnc8", "sed_gemm_qkv_lnc8"]


def build(per_gpu_batch, depth, device, mode="finetune2"):
    from copy import deepcopy
    from transformer4sed_amd import synth
    from transformer4sed_amd.passt_sed import PaSST_SED
    from transformer4sed_amd.scheduler import ExponentialDown
    from transformer4sed_amd.trainer import FusedAdamWEMA, MatSedTrainer, get_params
    cfg = MODE_CFG[mode]
    kw = dict(cfg["PaSST_SED"]["init_kwargs"])
    net = PaSST_SED(load_pretrained_model=False, encoder_depth=depth,
                    **{**kw, "passt_feature_layer": min(kw["passt_feature_layer"], depth)})
    sd = synth.matsed_state_dict_np(tag="w768", depth=12, mlm=bool(kw.get("mlm")))
    own = net.state_dict()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items() if k in own}, strict=True)
    net = net.to(device)
    ema_net = None
    if mode != "pretrain":
        ema_net = deepcopy(net)  # recipes/desed/finetune/passt/setting.py:8-15
        for p in ema_net.parameters():
            p.detach_()
    groups = get_params(net, cfg["opt"]["param_groups"])
    opt = FusedAdamWEMA(net, groups, ema_net=ema_net, betas=(0.9, 0.999), eps=1e-8)
    epoch_len = 1000
    sc = cfg["training"]["scheduler"]
    sched = ExponentialDown(opt, start_iter=sc["n_epochs_cut"] * epoch_len, total_iter=sc["n_epochs"] * epoch_len,
                            exponent=sc["exponent"], warmup_iter=sc["lr_warmup_epochs"] * epoch_len,
                            warmup_rate=sc["lr_warmup_rate"])
    net.train()
    if ema_net is not None:
        ema_net.train()  # the teacher runs in train mode during training (finetune/train.py:131-132)
    trainer = MatSedTrainer(net, ema_net, opt, sched, cfg, epoch_len)
    return net, ema_net, opt, trainer, sd


def build_pmam(depth, device):
    """PaSST_CNN + PmamTrainer as recipes/desed/pmam/main.py:89-171 wires them (synthetic weights and GMM prototypes)."""
    from transformer4sed_amd import synth
    from transformer4sed_amd.passt_cnn import PaSST_CNN
    from transformer4sed_amd.pmam_trainer import PmamTrainer, get_param_lr, mark_only_lora_as_trainable
    from transformer4sed_amd.scheduler import ExponentialDown
    from transformer4sed_amd.trainer import FusedAdamWEMA
    cfg = json.loads(json.dumps(PMAM))
    kw = cfg["PaSST_CNN"]["init_kwargs"]
    ps = dict(kw["passt_sed_param"], load_pretrained_model=False, encoder_depth=depth)
    ps["passt_feature_layer"] = min(ps["passt_feature_layer"], depth)
    net = PaSST_CNN(passt_sed_param=ps, cnn_param=kw["cnn_param"])
    sd = synth.pmam_state_dict_np(depth=12)
    own = net.state_dict()
    net.load_state_dict({k: torch.from_numpy(np.asarray(sd[k])) for k in own}, strict=True)
    net = net.to(device)
    mark_only_lora_as_trainable(net.backbone)
    groups = get_param_lr(net, cfg["opt"]["param_groups"])
    opt = FusedAdamWEMA(net, groups, ema_net=None, betas=(0.9, 0.999), eps=1e-8)
    epoch_len = 1000
    sc = cfg["training"]["scheduler"]
    sched = ExponentialDown(opt, start_iter=sc["n_epochs_cut"] * epoch_len, total_iter=sc["n_epochs"] * epoch_len, exponent=sc["exponent"],
                            warmup_iter=sc["lr_warmup_epochs"] * epoch_len, warmup_rate=sc["lr_warmup_rate"])
    gmm = torch.from_numpy(synth.det_normal("pmam/gmm_means", (30, 768)))
    net.train()
    return net, opt, PmamTrainer(net, opt, sched, gmm, cfg)


def build_dasm(depth, device, n_queries):
    """DASM (BASELINE.json config #5) as recipes/audioset_strong/detect_any_sound/passt/main.py:24 constructs it -- PaSST backbone, the
    PMAM CNN branch, 3-layer Transformer-XL SED decoder, 2-layer query decoder, 1024-wide external query embeddings (synthetic weights:
    the reference ships no YAML for this model; CLAP, which makes the embeddings, is not vendored)."""
    from transformer4sed_amd import synth
    from transformer4sed_amd.dasm import DASM
    cnn = {"n_in_channel": 1, "activation": "cg", "conv_dropout": 0.5, "kernel_size": [3] * 10, "padding": [1] * 10, "stride": [1] * 10,
           "nb_filters": list(synth.PMAM_FILTERS), "pooling": [list(p) for p in synth.PMAM_POOLING]}
    nb = max(1, n_queries // 2)
    net = DASM(cnn_param=cnn, backbone_param={"embed_dim": 768, "passt_feature_layer": min(10, depth), "pretrain_model_path": None, "lora_config": None},
               at_param={"at_decoder_layer": 2, "query_projector": True, "query_dim": 1024, "out_type": "sigmoid", "query": torch.zeros(nb, 1024)},
               decoder="transformerXL", decoder_layer_num=3, decoder_dim=768, num_heads=12, class_num=nb)
    if depth != 12:
        raise SystemExit("--mode dasm runs the reference's depth (12): DASM builds its PaSST with depth 12 (detect_any_sound.py:222)")
    sd = synth.dasm_full_state_dict_np(n_queries=nb, query_dim=1024)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    net = net.to(device).eval()
    g = torch.Generator().manual_seed(5)
    q = torch.nn.functional.normalize(torch.randn(n_queries, 1024, generator=g), dim=-1)
    q[:nb] = torch.from_numpy(sd["at_query"])
    mask = torch.ones(n_queries, n_queries, dtype=torch.bool)
    mask[:, :nb] = False
    mask.fill_diagonal_(False)
    return net, q.to(device), mask.to(device)


def build_dasm_train(depth, device, n_classes):
    """DASM + DasmTrainer for the closed-vocabulary training recipe: every class a learned query (`at_query`), whole model trainable, fused
    AdamW over the parameter groups of recipes/desed/finetune/cnn_trans/setting.py:get_param_lr (what the recipe family's main.py uses)."""
    from transformer4sed_amd import synth
    from transformer4sed_amd.dasm import DASM
    from transformer4sed_amd.dasm_trainer import DasmTrainer
    from transformer4sed_amd.pmam_trainer import get_param_lr
    from transformer4sed_amd.scheduler import ExponentialDown
    from transformer4sed_amd.trainer import FusedAdamWEMA
    cfg = json.loads(json.dumps(DASM_TRAIN))
    cnn = {"n_in_channel": 1, "activation": "cg", "conv_dropout": 0.5, "kernel_size": [3] * 10, "padding": [1] * 10, "stride": [1] * 10,
           "nb_filters": list(synth.PMAM_FILTERS), "pooling": [list(p) for p in synth.PMAM_POOLING]}
    sd = synth.dasm_full_state_dict_np(n_queries=n_classes, query_dim=1024)
    net = DASM(cnn_param=cnn, backbone_param={"embed_dim": 768, "passt_feature_layer": min(10, depth), "pretrain_model_path": None, "lora_config": None},
               at_param={"at_decoder_layer": 2, "query_projector": True, "query_dim": 1024, "out_type": "sigmoid",
                         "query": torch.from_numpy(np.asarray(sd["at_query"])).clone()},
               decoder="transformerXL", decoder_layer_num=3, decoder_dim=768, num_heads=12, class_num=n_classes, _encoder_depth=depth)
    # synthetic sed_head of a fifth the synthetic size (as the training fixtures, oracle/make_golden.py gen_dasm_train): frame logits of order 1
    sd["sed_head.weight"] = np.asarray(sd["sed_head.weight"]) * np.float32(0.2)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
    net = net.to(device)
    groups = get_param_lr(net, cfg["opt"]["param_groups"])
    opt = FusedAdamWEMA(net, groups, ema_net=None, betas=(0.9, 0.999), eps=1e-8)
    sc = cfg["training"]["scheduler"]
    epoch_len = 1000
    sched = ExponentialDown(opt, start_iter=sc["n_epochs_cut"] * epoch_len, total_iter=sc["n_epochs"] * epoch_len, exponent=sc["exponent"],
                            warmup_iter=sc["lr_warmup_epochs"] * epoch_len, warmup_rate=sc["lr_warmup_rate"])
    net.train()
    return net, opt, DasmTrainer(net, opt, sched, cfg, sr=16000)


def cpu_baseline(depth, batch=4, budget_s=330):
    """Oracle (torch CPU fp32) timed on the host cores in a child process with a hard time budget: the full finetune2 step
    (train-mode frontend, full augmentation, student fwd+bwd, 11-window teacher fwd, losses, AdamW, EMA) at batch `batch`,
    one warm-up step then the median of three timed steps on the same inputs."""
    import subprocess
    from transformer4sed_amd.hostcpu import usable_cpus
    cores = usable_cpus()           # affinity mask and cgroup CPU quota: more threads than that only buys throttling
    threads = min(cores, 64)
    cmd = [sys.executable, "-m", "oracle.cpu_step", str(batch), str(depth), str(threads), json.dumps(FINETUNE2)]
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS=str(threads))
    txt, note = "", ""
    try:
        txt = subprocess.run(cmd, capture_output=True, timeout=budget_s, env=env, text=True, cwd=ROOT).stdout
    except subprocess.TimeoutExpired as e:    # keep the steps that did finish
        txt = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
        note = f"; cut at the {budget_s} s budget"
    except Exception as e:  # the baseline leg must never take the GPU number down with it
        note = "; failed: " + repr(e)[:120]
    secs = []
    for ln in txt.splitlines():
        try:
            r = json.loads(ln)
            if not r["warmup"]:
                secs.append(r["sec"])
        except Exception:
            pass
    if not secs:
        return {"value": None, "unit": "clips/s", "cores": threads, "kind": "port", "sample": "no timed step finished" + note}
    med = float(np.median(secs))
    return {"value": round(batch / med, 4), "unit": "clips/s", "cores": threads, "kind": "port", "batch": batch,
            "sample": f"finetune2 step at batch {batch} (CPU oracle, torch fp32, full augmentation): 1 warm-up + median of {len(secs)} "
                      f"timed steps = {med:.1f} s/step, {threads} threads of {cores}" + note}


def run_pipe_leg(a, B, dev, rank, world, step, dist):
    """--mode pipe: N synthetic 16 kHz 16-bit RIFF files -> data.WavBatchStream (reader threads, pinned staging, one H2D copy and the
    polyphase resampler per batch on a side stream, `depth` batches in flight) -> the train step.  The same K steps as the resident leg,
    every step on a NEW batch of files; barrier + synchronize on both sides.  Replaces the reference's DataLoader (6 workers x
    librosa.load, src/preprocess/dataset.py:52-74, feats_extraction.py:7-38) over offline-resampled files (src/utils/resample.py)."""
    import shutil
    import tempfile
    from transformer4sed_amd import data, synth
    n_files = max(4 * B, 128)
    root = tempfile.mkdtemp(prefix=f"sed_pipe_{rank}_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        t_gen = time.perf_counter()
        base = synth.synth_wav(8, seed=4242 + rank)[:, ::2]            # 8 distinct 10 s clips at 16 kHz
        paths = []
        for i in range(n_files):
            x = np.roll(base[i % 8], 997 * i) * (0.5 + 0.5 * ((i * 7919) % 11) / 10.0)
            if i % 9 == 4:
                x = x[: 16000 * (2 + i % 7)]                            # shorter files take the zero-pad path
            pth = os.path.join(root, f"clip_{i:05d}.wav")
            data.write_wav(pth, x, 16000)
            paths.append(pth)
        t_gen = time.perf_counter() - t_gen
        nb = a.warmup + a.steps
        order = np.random.RandomState(77 + rank).permutation(n_files)
        batches = [[int(order[(k * B + j) % n_files]) for j in range(B)] for k in range(nb)]
        stream = data.WavBatchStream(paths, batches, dev, depth=a.pipe_depth, workers=a.pipe_workers)
        it = iter(stream)
        for _ in range(a.warmup):
            step(next(it)[0])
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        cpu0, t0 = time.process_time(), time.perf_counter()
        for _ in range(a.steps):
            out = step(next(it)[0])
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        cpu_s = time.process_time() - cpu0
        for _ in it:
            pass
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        if not np.isfinite(float(out["loss_total"])):
            raise SystemExit("non-finite loss in the pipe leg")
        return {"value": round(a.steps * B * world / dt, 3), "ms_per_step": round(1000 * dt / a.steps, 3), "files": n_files,
                "file_format": "RIFF PCM-16 mono 16 kHz, 10 s (1 in 9 shorter)", "file_bytes": 320044, "staging": root.split("/")[1],
                "reader_threads": a.pipe_workers, "batches_in_flight": a.pipe_depth, "h2d_bytes_per_clip": 2 * stream.L + 4,
                "cpu_ms_per_step": round(1000 * cpu_s / a.steps, 2), "file_generation_s": round(t_gen, 2)}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default 32; 24 for --mode pmam, the reference's batch)")
    ap.add_argument("--depth", type=int, default=12)
    ap.add_argument("--mode", default="finetune2", choices=["finetune2", "finetune1", "pretrain", "val", "pmam", "pipe", "dasm", "dasm_train"],
                    help="finetune2 = the headline train step (default); finetune1 / pretrain = the other two training stages of "
                         "the MAT-SED recipe; val = Trainer.validation's per-batch body (student + teacher, 17 sliding windows, "
                         "score tables + event decoding), SURVEY 8(f) rank 1; pmam = the PMAM post-pretrain step (PaSST_CNN, SURVEY 8(f) rank 3)")
    ap.add_argument("--pipe-step", default="finetune2", choices=["finetune2", "pretrain"],
                    help="--mode pipe: which train step consumes the file stream (synthetic 16 kHz RIFF files -> data.WavBatchStream -> "
                         "device resampler -> step); the line reports end-to-end clips/s beside the resident-input rate of the same process")
    ap.add_argument("--dasm-queries", type=int, default=None,
                    help="--mode dasm: number of query embeddings per call (half of them 'base' queries, half novel ones behind the "
                         "open-vocabulary attention mask; default 64); --mode dasm_train: number of classes = learned queries "
                         "(default 407 = every AudioSet-strong class, the configuration the step's FLOP figure was counted on)")
    ap.add_argument("--pipe-workers", type=int, default=2)
    ap.add_argument("--pipe-depth", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    a = ap.parse_args()

    pipe = a.mode == "pipe"
    if pipe:
        a.mode = a.pipe_step      # everything below builds and runs that step; only the input source of the timed loop differs
    one_gpu = os.environ.get("SED_BENCH_ONE_GPU") == "1"
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # `python bench.py --gpus N` on its own: become N ranks (one process per GPU) through torch.distributed.run, exactly the
        # command the driver would have used; the children take the branch below
        have = torch.cuda.device_count()
        if have < a.gpus and not one_gpu:
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {have} HIP device(s) visible (one rank per GPU; nothing was measured)")
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} was started with WORLD_SIZE={world}: launch one rank per GPU "
                         f"(torch.distributed.run --nproc-per-node {a.gpus}) or drop the launcher and let --gpus start the ranks")
    if world > 1 and not one_gpu and torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} HIP device(s) visible")
    force_ddp = os.environ.get("SED_DDP_FORCE") == "1"   # exercise the RCCL gradient path on a single rank (debugging aid)
    if world > 1 or force_ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # debugging aid for boxes with a single GPU: SED_BENCH_BACKEND=gloo SED_BENCH_ONE_GPU=1 runs every rank on cuda:0 with
        # host-staged collectives, which exercises the whole N > 1 control flow (RCCL itself refuses two ranks on one device)
        backend = os.environ.get("SED_BENCH_BACKEND", "nccl")
        if one_gpu:
            local = 0           # (LOCAL_RANK still decides who builds)
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
        if dist.get_world_size() != world:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, expected {world}")
        world = dist.get_world_size()      # n_gpus / global_batch below come from the LIVE group, not from the flag
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    # the product library only: the benchmark process never imports the checker (oracle/); the cpu_baseline leg runs it in a child
    from transformer4sed_amd import build as _build, hostcpu
    from transformer4sed_amd import _lib as _L
    if world > 1:
        # one builder per node: concurrent hipcc runs writing the same libsed_hip.so would corrupt it
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            _build.build(verbose=False)
        dist.barrier()
    _build.build(verbose=False)   # up to date by now
    _L.lib()                      # dlopen + every symbol of include/sed_hip.h
    hostcpu.cap_torch_threads()   # entry points opt in to the host-thread cap; importing the package does not
    from transformer4sed_amd import ops, synth
    from transformer4sed_amd.ddp import GradBucketReducer
    import random
    random.seed(1000 + rank); np.random.seed(1000 + rank); torch.manual_seed(1000 + rank)

    if a.dasm_queries is None:
        a.dasm_queries = 407 if a.mode == "dasm_train" else 64
    B = a.batch or (24 if a.mode in ("pmam", "dasm", "dasm_train") else 32)
    if a.mode == "dasm_train":
        if a.depth != 12:
            raise SystemExit("--mode dasm_train runs the reference's depth (12): DASM builds its PaSST with depth 12 (detect_any_sound.py:213)")
        net, opt, trainer = build_dasm_train(a.depth, dev, a.dasm_queries)
        ema_net = None
    elif a.mode == "dasm":
        net, dasm_q, dasm_mask = build_dasm(a.depth, dev, a.dasm_queries)
        ema_net = opt = trainer = None
    elif a.mode == "pmam":
        net, opt, trainer = build_pmam(a.depth, dev)
        ema_net = None
    else:
        net, ema_net, opt, trainer, sd = build(B, a.depth, dev, a.mode)
    # batch composition strong+synth | weak | unlabeled in the reference's positional order (dataset.py:178-188)
    sn = (B * 4 + 11) // 12
    wn = (B * 4 + 11) // 12
    un = B - sn - wn
    if trainer is not None:
        trainer.cfg = json.loads(json.dumps(MODE_CFG[a.mode]))
    if a.mode == "dasm_train":
        trainer.config = trainer.cfg
    if a.mode not in ("pretrain", "pmam", "dasm", "dasm_train"):
        trainer.cfg["training"]["batch_size"] = [sn, 0, wn, un]
    wav = torch.from_numpy(synth.synth_wav(B, seed=1000 + rank)).to(dev)
    if a.mode == "pmam":   # frame-wise pseudo labels over the 30 GMM prototypes (FrameWiseLabeledDataset, pmam/setting.py:47-70)
        labels = torch.from_numpy(synth.synth_strong_labels(B, n_classes=30, seed=1000 + rank)).to(dev)
    elif a.mode == "dasm_train":   # strong labels over the query classes (StronglyLabeledDataset, recipes/audioset_strong/setting.py:161-166)
        labels = torch.from_numpy(synth.synth_strong_labels(B, n_classes=a.dasm_queries, seed=1000 + rank)).to(dev)
    else:
        labels = torch.from_numpy(synth.synth_batch_labels(sn, wn, un, seed=1000 + rank)).to(dev)
    if (world > 1 or force_ddp) and trainer is not None:
        trainer.ddp = GradBucketReducer(net, opt)
        trainer.ddp.force = force_ddp

    finish = None       # mode-specific work that belongs to the timed steps but runs once behind them
    if a.mode == "val":
        from transformer4sed_amd.evaluation import Encoder, Evaluator
        enc = Encoder(["Alarm_bell_ringing", "Blender", "Cat", "Dishes", "Dog", "Electric_shaver_toothbrush", "Frying",
                       "Running_water", "Speech", "Vacuum_cleaner"], audio_len=10, frame_len=1024, frame_hop=320, net_pooling=1,
                      sr=32000)
        vcfg = {"training": {"median_window": [5, 20, 5, 5, 5, 20, 20, 20, 5, 20], "filter_type": "median", "weak_mask": True},
                "PaSST_SED": {"val_kwargs": {"encoder_win": True, "win_param": [512, 31], "mix_rate": 0.5, "temp_w": 0.5}}}
        pad_mask = torch.zeros(B, 1000, dtype=torch.bool)
        paths = [f"/synthetic/val/clip_{rank}_{i}.wav" for i in range(B)]

        # one Evaluator for the whole run, like Trainer.validation over an epoch: the host half of a batch's decode runs under the next
        # forward; `finish` (inside the timed region) completes the last batch's tables
        ev = Evaluator(net, ema_net, enc, vcfg)

        def step(w=None):
            ev.step(wav, labels, pad_mask, paths)
            return {"loss_total": torch.tensor(float(B))}

        def finish():
            ev.flush()
            assert len(ev.scores.post_student) == B and len(ev.scores.post_teacher) == B
    elif a.mode == "dasm":
        ext = net.get_feature_extractor()

        def step(w=None):      # open-vocabulary inference: eval frontend -> whole model -> frame posteriors of every query
            with torch.no_grad():
                mel = ext.logmel(wav if w is None else w)
                strong, weak, other = net(mel, temp_w=0.5, query=dasm_q, tgt_mask=dasm_mask)
            return {"loss_total": weak.mean()}
    elif a.mode == "pretrain":
        def step(w=None):
            out = trainer.pretrain_step(wav if w is None else w)
            return {"loss_total": out["loss"]}
    elif a.mode in ("pmam", "dasm_train"):
        def step(w=None):
            return trainer.step(wav, labels.clone())
    else:
        def step(w=None):
            return trainer.finetune_step(wav if w is None else w, labels.clone())

    for _ in range(a.warmup):
        step()
    if os.environ.get("SED_SYNC_DEBUG"):      # developer switch: warn (with a stack) at every host synchronisation inside the steps
        torch.cuda.set_sync_debug_mode(1)
    # HIP events bracket every GEMM launch of the FIRST timed step only: the event packets cost ~8 us of stream time per launch
    # (2.5 ms on a 138 ms step when every step is instrumented), which would otherwise be charged to `value`
    timer = None if a.no_kernel_timer else ops.KernelTimer(GEMM_KERNELS + list(ops.HBM_KERNELS) + (["sed_gemm_f32", "sed_xattn_f32_fwd", "sed_xattn_f32_fwd_train", "sed_xattn_f32_bwd"] if a.mode in ("dasm", "dasm_train") else []))
    timed_steps_with_events = 1
    if timer is not None:       # one untimed instrumented step creates the event objects; the timed step reuses them
        ops.TIMER = timer
        step()
        ops.TIMER = None
        torch.cuda.synchronize()
        timer.recycle()
    from transformer4sed_amd.gpumon import GpuSampler
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    mon = GpuSampler(local).start() if rank == 0 else None
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    for i in range(a.steps):
        ops.TIMER = timer if i < timed_steps_with_events else None
        out = step()
    if finish is not None:
        finish()
    host_issue_s = time.perf_counter() - t0      # the Python schedule of the timed steps has been issued (the GPU is still running them)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cpu_s = time.process_time() - cpu0
    gpu_state = mon.stop() if mon is not None else None
    ops.TIMER = None
    pipe_info = None
    if pipe:
        pipe_info = run_pipe_leg(a, B, dev, rank, world, step, dist)
    # one more UNTIMED instrumented step with the LayerNorm fold switched off: the GEMM family on its own, for the roofline note (the
    # timed steps above ran the product's default, where the no-grad GEMM launches also carry their block's LayerNorm)
    summ_unfolded = None
    if world == 1 and timer is not None and a.mode in ("finetune2", "finetune1", "pretrain"):      # (a step is collective: single rank only)
        engs = [e for e in (getattr(net, "engine", None), getattr(getattr(trainer, "ema_net", None), "engine", None)) if e is not None]
        if engs and all(getattr(e, "ln_fold", False) for e in engs):
            t2 = ops.KernelTimer(GEMM_KERNELS)
            for e in engs:
                e.ln_fold = False
            try:
                ops.TIMER = t2
                step()
                ops.TIMER = None
                torch.cuda.synchronize()
                summ_unfolded = t2.summarize()
            finally:
                ops.TIMER = None
                for e in engs:
                    e.ln_fold = True
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(out["loss_total"])
    if not np.isfinite(loss):
        raise SystemExit("non-finite loss in the timed region")
    clips = a.steps * B * world
    value = clips / dt
    gflop_clip, gflop_batch = MODE_GFLOP[a.mode]
    if a.mode == "dasm_train" and a.dasm_queries != 407:
        gflop_clip = gflop_batch = None      # (the reference's FLOPs were counted with 407 queries)
    line = {
        "metric": {"finetune2": "clips/sec (10 s clips) MAT-SED finetune2 train step",
                   "finetune1": "clips/sec (10 s clips) MAT-SED finetune1 train step",
                   "pretrain": "clips/sec (10 s clips) MAT-SED masked-reconstruction pretrain step",
                   "val": "clips/sec (10 s clips) MAT-SED validation step (student + teacher, 17 windows, score tables + events)",
                   "pmam": "clips/sec (10 s clips) PMAM post-pretrain step (PaSST_CNN: LoRA encoder + CNN branch, prototype BCE)",
                   "dasm": "clips/sec (10 s clips) DASM open-vocabulary inference (frontend + PaSST + CNN + SED decoder + query decoder + dual-stream head)",
                   "dasm_train": "clips/sec (10 s clips) DASM train step (DASMTrainer.train: frontend + augmentation, forward, BCE on frame posteriors + tagging, backward, AdamW)"}[a.mode],
        "value": round(value, 3), "unit": "clips/s",
        "n_gpus": world, "ranks": dist.get_world_size() if dist.is_initialized() else 1,
        "collective_backend": (dist.get_backend() if dist.is_initialized() else None), "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1000 * dt / a.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 fwd / bf16 bwd MFMA operands, fp32 accumulate + residual stream",
        "data": "synthetic (DESED-shaped 320000-sample clips @32 kHz, deterministic synthetic weights)",
        "config": {"workload": "MAT-SED base finetune2 mean-teacher step (config/mat-sed/base/finetune2.yaml), student fwd+bwd "
                               "all-trainable + teacher fwd with 11 sliding windows + AdamW + EMA",
                   "model": f"PaSST_SED depth {a.depth} + 3x TransformerXL context net (100.95 M params)",
                   "global_batch": B * world, "per_gpu_batch": B, "seq_len": "1190 encoder tokens / 1000 decoder frames",
                   "parallelism": f"dp{world}", "final_loss": loss},
    }
    line["gpu_state"] = gpu_state
    line["host"] = {"cpus_usable": hostcpu.usable_cpus(), "affinity": len(os.sched_getaffinity(0)),
                    "issue_ms_per_step": round(1000 * host_issue_s / a.steps, 2), "cpu_ms_per_step": round(1000 * cpu_s / a.steps, 2),
                    "note": "issue = wall time until the Python schedule of the timed steps was issued (includes the back-pressure of the launch "
                            "queue once it is several steps deep: an upper bound of the host's own time); cpu = process CPU time (all threads) per step"}
    if pipe_info is not None:
        # the headline of this mode is the END-TO-END rate; the resident-input rate measured above in the same process sits beside it
        line["pipe"] = dict(pipe_info, resident_value=line["value"], resident_ms_per_step=line["ms_per_step"],
                            ratio_vs_resident=round(pipe_info["value"] / line["value"], 4))
        line["metric"] = line["metric"] + ", inputs streamed from 16 kHz RIFF files (read + H2D + device resampler inside the timed region)"
        line["value"], line["ms_per_step"] = pipe_info["value"], pipe_info["ms_per_step"]
        line["data"] = "synthetic 16-bit PCM 16 kHz files (DESED-shaped 10 s clips, some shorter), deterministic synthetic weights"
    if getattr(trainer, "ddp", None) is not None:
        st = trainer.ddp.last_stats
        line["grad_exchange"] = {"dtype": str(trainer.ddp.comm_dtype).replace("torch.", ""), "collectives_per_step": st["collectives"],
                                 "MB_per_step": round(st["bytes"] / 1e6, 1), "reserved_cus": trainer.ddp.reserve_cus,
                                 "note": "stage-triggered all-reduce(AVG) of contiguous gradient-arena slices on the collective stream while "
                                         "the backward continues (ddp.py); SED_DDP_COMM_DTYPE=bf16 halves the bytes"}
    if gflop_clip is not None:
        # fraction of the dense 16-bit MFMA peak over the whole step, on the FLOPs this build executes (algorithmic: no credit for the
        # 3x K of the split-precision GEMMs or for padded columns); the same on the reference's own schedule is reported beside it
        skipped = GFLOP_SKIPPED.get(a.mode, 0.0)
        line["step_mfma_frac"] = round(value / world * (gflop_clip - skipped + gflop_batch / B) / 1000.0 / PEAK_BF16_TFLOPS, 4)
        line["step_gflop_per_clip"] = {"executed": round(gflop_clip - skipped, 1), "reference_schedule": gflop_clip,
                                       "skipped": (("blocks 11-12 of the 17 windows of student and teacher" if a.mode == "val" else "blocks 11-12 of the 11 teacher windows") + " (their output is never read)") if skipped else None}
        line["step_mfma_frac_reference_flops"] = round(value / world * (gflop_clip + gflop_batch / B) / 1000.0 / PEAK_BF16_TFLOPS, 4)
    if rank == 0 and timer is not None:
        summ_all = timer.summarize()
        summ = {k: v for k, v in summ_all.items() if k in GEMM_KERNELS}
        # HBM-bound kernels of the same instrumented step (SURVEY 8(d) "report both"): algorithmic bytes / HIP-event duration vs 8 TB/s
        what = {"sed_logmel_fwd": "log-mel frontend (wav_absmax + logmel kernels; 1.28 MB read + 0.512 MB written per clip)",
                "sed_adamw_ema": "fused AdamW + EMA sweeps (28 B / trainable parameter + 8-12 B / EMA parameter)",
                "sed_layernorm_fwd": "LayerNorm forward (fp32 in -> 16-bit / fp32 out + statistics)",
                "sed_layernorm_bwd": "LayerNorm backward (dy, x in; dx out / accumulated; gamma / beta gradients)",
                "sed_layernorm_bwd_x16": "LayerNorm backward that also writes the bf16 image of the gradient stream (the next GEMMs' dY operand)"}
        line["roofline_hbm"] = [
            {"kernel": k, "what": what[k], "launches": v["launches"], "bytes": round(v["bytes"]), "us": round(1000 * v["ms"], 1),
             "GB/s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1), "peak_GB/s": PEAK_HBM_GBS,
             "frac_of_8TB/s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
             # (beside the data-sheet figure: what a pure streaming kernel reaches on this part, MI355X_MICROARCH.md's measured HBM rate)
             "frac_of_measured_6.3TB/s": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / MEASURED_HBM_GBS, 4)}
            for k, v in summ_all.items() if k in what and v["ms"] > 0]
        if a.mode in ("dasm", "dasm_train") and "sed_gemm_f32" in summ_all:
            v = summ_all["sed_gemm_f32"]
            xs_ = [summ_all[k] for k in ("sed_xattn_f32_fwd", "sed_xattn_f32_fwd_train", "sed_xattn_f32_bwd") if k in summ_all]
            x = {"ms": sum(t["ms"] for t in xs_), "launches": sum(t["launches"] for t in xs_), "flops": sum(t["flops"] for t in xs_)}
            line["roofline_f32"] = {"kernel": "gemm_f32_kernel (query decoder / head linears and their dX / dW products, folded memory projection, per-clip einsum) on v_mfma_f32_32x32x2_f32",
                                    "bound": "mfma", "achieved": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s",
                                    "frac": round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / 157.3, 4), "launches": v["launches"], "ms": round(v["ms"], 3),
                                    "xattn_f32_ms": round(x["ms"], 3), "xattn_f32_launches": x["launches"],
                                    "xattn_f32_TFLOP/s": round(x["flops"] / (x["ms"] * 1e-3) / 1e12, 2) if x["ms"] > 0 else None,
                                    "xattn_note": "sed_xattn_f32_* (fp32 in / out; the name is the interface): every product as a three-term split-precision product on the "
                                                  "16-bit matrix pipe (f16 pairs for q, k, v, P; bf16 pairs for gradient operands; fp32 accumulation) -- algorithmic FLOPs "
                                                  "here, 3x as many issued; NOT priced against the fp32-MFMA peak of this block",
                                    "note": "fp32-input MFMA peak = the fp32 vector peak (MI355X_MICROARCH.md); exact fp32 products and accumulation (GEMMs of this block)"}
        ms = sum(v["ms"] for v in summ.values())
        fl = sum(v["flops"] for v in summ.values())
        n = sum(v["launches"] for v in summ.values())
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        fl_issued = sum(v["flops_issued"] for v in summ.values())
        by = sum(v["bytes"] for v in summ.values())
        # PMC figures (separate rocprofv3 passes over this command, tools/gemm_traffic.py / tools/mfma_util.py) are keyed by mode:
        # a mode without a committed profile reports null, never another workload's numbers
        # ... and only for the configuration the profile was taken on (depth 12, the mode's default per-GPU batch)
        def prof(kind):
            if a.depth != 12 or a.batch is not None:
                return None, None
            for rnd in ("r6", "r5", "r4", "r3", "r2"):
                path = os.path.join(ROOT, "profiles", f"{rnd}_gemm_{kind}_{a.mode}.json")
                if os.path.exists(path):
                    return json.load(open(path)), os.path.relpath(path, ROOT)
            return None, None
        tj, tsrc = prof("traffic")
        mj, msrc = prof("mfma_busy")
        line["roofline"] = {"kernel": "gemm_nt_pp_kernel<EPI,F16> / gemm_tn_dw_kernel / gemm_nt_kernel (all GEMM launches of the step: "
                                      "linears, qkv, dX, dW)",
                            "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                            "achieved_issued": round(fl_issued / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, 2),
                            "achieved_note": "achieved = algorithmic 2MNK (split-precision GEMMs counted at their logical K); "
                                             "achieved_issued = MFMA FLOPs actually issued (3x K for the split-precision GEMMs, 2x K for the two-term-weight GEMMs of evaluation passes -- for sed_gemm_*_w2f8 the second K pass is e4m3 x e4m3 on the fp8 matrix path, counted here at its FLOPs); "
                                             "the GEMM launches of no-grad encoder passes also carry that block's LayerNorm (row statistics in the residual epilogue, "
                                             "normalisation in the next GEMM's epilogue; SED_LN_FOLD=0 separates them again): their time counts here, the LayerNorm passes "
                                             "they replace do not exist any more",
                            "traffic": None if tj is None else tj.get("avg_bytes_per_launch"), "traffic_unit": "HBM bytes per launch",
                            "traffic_source": tsrc, "mfma_pipe_busy": None if mj is None else mj.get("family_busy_fraction"),
                            "mfma_pipe_busy_source": msrc,
                            "pmc_collected_on_commit": None if tj is None else tj.get("commit"),
                            "achieved_ln_unfolded": None if not summ_unfolded else round(
                                sum(v["flops"] for v in summ_unfolded.values()) / (sum(v["ms"] for v in summ_unfolded.values()) * 1e-3) / 1e12, 2),
                            "achieved_ln_unfolded_note": "the same GEMM launches in one extra untimed step with SED_LN_FOLD=0 (LayerNorm as separate kernels): "
                                                         "the family's own rate; `achieved` / `frac` are the timed configuration",
                            "alg_bytes_per_launch": round(by / max(1, n)),
                            "launches_per_step": n // timed_steps_with_events, "avg_launch_ms": round(ms / max(1, n), 4),
                            "instrumented_steps": f"{timed_steps_with_events} of the {a.steps} timed steps",
                            "gemm_share_of_step": round(ms / timed_steps_with_events / (1000 * dt / a.steps), 3),
                            "flops_per_launch": round(fl / max(1, n) / 1e9, 3)}
        clk = ((gpu_state or {}).get("sclk_MHz") or {}).get("mean")
        if clk:      # the same rate against what the matrix pipes can do at the clock the step actually ran at (the chip clocks to its power budget)
            line["roofline"]["sclk_MHz_mean"] = clk
            line["roofline"]["frac_at_measured_clock"] = round(ach / (PEAK_BF16_TFLOPS * clk / 2400.0), 4)
    if a.mode in ("finetune1", "pretrain"):
        line["config"]["workload"] = {"finetune1": "MAT-SED base finetune1 step (config/mat-sed/base/finetune1.yaml): encoder and context "
                                                   "net frozen, heads trained, mean-teacher losses, teacher without windows",
                                      "pretrain": "MAT-SED base pretrain step (config/mat-sed/base/pretrain.yaml): masked-frame "
                                                  "reconstruction (75 % block mask), encoder frozen, context net + MLM head trained"}[a.mode]
    if a.mode == "pmam":
        line["config"]["workload"] = ("PMAM post-pretrain step (config/pmam/post_pretrain.yaml): PaSST encoder with LoRA r=8 (blocks 9-12 "
                                      "trainable), 10-layer CNN branch, attention frequency pooling, 384-wide context net, 80 % block mask, "
                                      "prototype-similarity BCE + 0.1 AT BCE, AdamW")
        line["config"]["model"] = f"PaSST_CNN depth {a.depth} (97.8 M params, 11.7 M trainable)"
    if a.mode == "dasm":
        line["config"]["workload"] = (f"DASM inference batch (src/models/detect_any_sound/detect_any_sound.py:324-399): eval frontend, PaSST depth 12 + "
                                      f"10-layer CNN branch, attention pooling, 3-layer Transformer-XL SED decoder, 2-layer query decoder over the 1188 "
                                      f"patch tokens, {a.dasm_queries} query embeddings ({max(1, a.dasm_queries // 2)} base + novel ones behind the "
                                      f"open-vocabulary attention mask), temperature 0.5; forward only")
        line["config"]["model"] = "DASM depth 12 (PaSST + CNN + Transformer-XL + query decoder, 119.7 M params), synthetic weights and query embeddings"
        line["config"].pop("final_loss", None)
        line["dtype"] = ("f16 MFMA operands in the encoder / SED decoder (split precision there); query decoder and head: fp32-input MFMA GEMMs, attention as "
                         "three-term f16-pair products with fp32 accumulation")
    if a.mode == "dasm_train":
        line["config"]["workload"] = (f"DASM train step (recipes/audioset_strong/detect_any_sound/passt/train.py:66-120): train-mode frontend, frame_shift / "
                                      f"mixup / FilterAugment, PaSST depth 12 + CNN branch + Transformer-XL SED decoder + 2-layer query decoder (dropout 0.1) over "
                                      f"the 1188 patch tokens with {a.dasm_queries} learned class queries, BCE on [B, {a.dasm_queries}, 1000] frame posteriors + 0.5 x "
                                      f"BCE on the tagging probabilities, backward through everything (whole model trainable), fused AdamW")
        line["config"]["model"] = "DASM depth 12 (PaSST + CNN + Transformer-XL + query decoder), synthetic weights and query embeddings"
        line["dtype"] = ("f16 fwd / bf16 bwd MFMA operands in the encoder / CNN / SED decoder (split precision there); query decoder and dual-stream head: "
                         "fp32-input MFMA GEMMs (16-bit split precision for Linears with >= 1024 rows), attention forward and backward as three-term "
                         "split-precision products (f16 pairs for q, k, v, P; bf16 pairs for gradient operands), fp32 accumulation")
    if a.mode == "val":
        line["config"]["workload"] = ("MAT-SED validation batch (recipes/desed/finetune/train.py:296-366): eval frontend, student and "
                                      "EMA teacher forward with val_kwargs (17 windows of 512 frames, step 31, temp 0.5), soft-masked "
                                      "scipy-median score tables and half-point event decoding")
        line["config"].pop("final_loss", None)
        w2 = getattr(getattr(net, "engine", None), "w2_f8_set", None) if getattr(getattr(net, "engine", None), "w2_f8", False) else None
        line["dtype"] = ("evaluation-mode encoder on two-term f16 weights [f16(W) | W - f16(W)]: hi products f16 x f16 MFMA, lo products " +
                         (f"e4m3 x e4m3 on the fp8 matrix path for {sorted(w2)} (SED_ENC_W2), f16 x f16 for the other GEMMs" if w2 else "f16 x f16") +
                         "; fp32 accumulate + residual stream; context network in split precision")
    if rank == 0 and world == 1 and not a.no_cpu_baseline and a.mode == "finetune2" and not pipe:      # (N = 1 only: the other ranks would sit in the final barrier meanwhile)
        line["cpu_baseline"] = cpu_baseline(a.depth)
    # RCCL writes its version banner through C stdio (block-buffered when piped): every rank pushes it out before the last
    # barrier so that rank 0's JSON line is the last thing on the job's stdout
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if world > 1 or force_ddp:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
