"""Multi-process (gloo, world_size 2, CPU) test of the data-parallel gradient exchange: the bucketed, stage-triggered
all-reduce of the flat gradient arena must give every rank the mean gradient -- i.e. what a single process would compute
on the concatenated batch for mean-reduced losses."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeOpt:
    def __init__(self, layout):
        self.layout = layout
        self.total = max(o + (k + 63) // 64 * 64 for _, o, k in layout)


class _FakeNet:
    depth = 2


NAMES = ["backbone.blocks.0.attn.qkv.weight", "backbone.blocks.0.mlp.fc1.weight", "backbone.blocks.1.attn.qkv.weight",
         "backbone.norm.weight", "backbone.patch_embed.proj.weight", "decoder.encoder_blocks.0.attn.in_proj.weight",
         "classifier.weight", "out_norm.weight", "at_adpater.1.weight",
         # PMAM variant (PaSST_CNN): LoRA factors ride with their block, everything else with the last stage
         "backbone.blocks.1.attn.qkv.lora_A", "cnn.cnn.conv0.weight", "cnn.cnn.batchnorm3.weight", "f_pool_module.f_att_token",
         "transformer_projector.weight", "merge_weight", "mask_token", "mlm_mlp.2.weight"]


def _layout():
    out, off = [], 0
    for i, n in enumerate(NAMES):
        k = 100 + 37 * i
        out.append((n, off, k))
        off += (k + 63) // 64 * 64
    return out


def _worker(rank, world, port, frozen_encoder):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transformer4sed_amd.ddp import GradBucketReducer
    net, opt = _FakeNet(), _FakeOpt(_layout())
    red = GradBucketReducer(net, opt, min_bytes=0)
    torch.manual_seed(rank)
    arena = torch.randn(opt.total)
    mine = arena.clone()
    net._last_grad_arena = arena
    # the engine fires these as the backward proceeds (engine.SedEngine.backward)
    red.on_stage("decoder")
    red.on_stage("heads")
    if not frozen_encoder:
        for li in (1, 0):
            red.on_stage(("block", li))
        red.on_stage("embed")
    red.allreduce_grads()
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    want = sum(gathered) / world
    assert torch.allclose(arena, want, atol=1e-6), (rank, float((arena - want).abs().max()))
    dist.destroy_process_group()


@pytest.mark.parametrize("frozen_encoder", [False, True])
def test_bucketed_allreduce_mean_world2(frozen_encoder):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, frozen_encoder), nprocs=2, join=True)


def test_stage_partition_covers_arena_once():
    from transformer4sed_amd.ddp import GradBucketReducer
    net, opt = _FakeNet(), _FakeOpt(_layout())
    red = GradBucketReducer(net, opt, min_bytes=0)
    covered = sorted(r for rs in red.ranges.values() for r in map(tuple, rs))
    pos = 0
    for a, b in covered:
        assert a == pos
        pos = b
    assert pos == opt.total


def test_param_groups_and_optimizer_runs_cpu():
    """get_params mirrors recipes/desed/finetune/passt/setting.py (step_lr groups, freezing); the optimiser only
    sweeps parameters that received a gradient, in contiguous runs."""
    from transformer4sed_amd.passt_sed import PaSST_SED
    from transformer4sed_amd.trainer import FusedAdamWEMA, get_params
    net = PaSST_SED(decoder="transformerXL", decoder_layer_num=3, at_adapter=True, load_pretrained_model=False,
                    encoder_depth=12)
    cfg2 = {"encoder": {"lr": 5e-6, "weight_decay": 1e-4, "freeze_layer": 0, "step_lr": 4},
            "decoder": {"lr": 1e-4, "weight_decay": 1e-4}, "head": {"lr": 1e-4, "weight_decay": 1e-4}}
    groups = get_params(net, cfg2)
    assert [g["lr"] for g in groups] == [5e-6, 1e-5, 1e-4, 1e-4]
    high = {n for n, _ in groups[1]["params"]}
    assert "backbone.blocks.8.attn.qkv.weight" in high and "backbone.norm.weight" in high
    assert "backbone.blocks.7.attn.qkv.weight" not in high and "backbone.blocks.3.norm1.weight" not in high  # only "norm." (final norm)
    assert all(p.requires_grad for p in net.parameters())
    n_all = sum(p.numel() for p in net.parameters())
    assert n_all == 100947762  # 100.95 M parameters
    cfg1 = {"encoder": {"lr": 0, "weight_decay": 1e-4, "freeze_layer": 0, "step_lr": 4},
            "decoder": {"lr": 0, "weight_decay": 1e-4}, "head": {"lr": 2e-4, "weight_decay": 1e-4}}
    net1 = PaSST_SED(decoder="transformerXL", decoder_layer_num=3, at_adapter=True, load_pretrained_model=False)
    get_params(net1, cfg1)
    trainable = sum(p.numel() for p in net1.parameters() if p.requires_grad)
    # finetune1 trains out_norm, at_adpater, classifier, backbone.norm only (2.38 M + the dead backbone.head)
    assert not net1.backbone.blocks[0].attn.qkv.weight.requires_grad and net1.backbone.norm.weight.requires_grad
    assert not net1.decoder.encoder_blocks[0].mlp.fc1.weight.requires_grad and net1.out_norm.weight.requires_grad
    opt = FusedAdamWEMA(net, groups)
    names = opt.param_groups[3]["names"]
    runs = opt._runs(names, set(names[:2]) | set(names[3:5]))
    assert len(runs) == 2 and all(a % 64 == 0 and b % 64 == 0 for a, b in runs)
    assert net._flat_layout is opt and opt.arena.numel() == opt.total
    assert net.classifier.weight.data_ptr() == opt.arena[opt.offset["classifier.weight"][0]:].data_ptr()
