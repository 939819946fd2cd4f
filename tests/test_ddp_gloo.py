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
         "transformer_projector.weight", "merge_weight", "mask_token", "mlm_mlp.2.weight",
         # DASM (row (g)): the query decoder and the dual-stream head finish their backward before the SED decoder's -- "decoder" stage;
         # norm_after_merge sits between the trunk and the head -- "heads"
         "at_decoder.layers.0.self_attn.in_proj_weight", "at_decoder.layers.1.multihead_attn.out_proj.weight", "at_projector.weight",
         "query_projector.0.weight", "at_query", "mask_embedding_layer.0.weight", "sed_head.weight", "at_head.bias", "norm_after_merge.weight"]


def _layout():
    out, off = [], 0
    for i, n in enumerate(NAMES):
        k = 100 + 37 * i
        out.append((n, off, k))
        off += (k + 63) // 64 * 64
    return out


def _worker(rank, world, port, frozen_encoder, comm_dtype=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transformer4sed_amd.ddp import GradBucketReducer
    net, opt = _FakeNet(), _FakeOpt(_layout())
    red = GradBucketReducer(net, opt, min_bytes=0, comm_dtype=comm_dtype)
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
    if comm_dtype == torch.bfloat16:
        # the exchange carries bf16 images: every rank's slice is rounded once (2^-9 relative), the mean once more; untouched padding /
        # excluded slices keep their fp32 values, and every rank ends with the SAME numbers (bit-identical replicas)
        img = sum(g.to(torch.bfloat16).float() for g in gathered) / world
        err = (arena - img).abs()
        assert float((err / (img.abs() + 1e-3)).max()) < 2 ** -7, float((err / (img.abs() + 1e-3)).max())
        assert float((arena - want).abs().max()) < 2 ** -6 * float(want.abs().max())
        both = [torch.zeros_like(arena) for _ in range(world)]
        dist.all_gather(both, arena)
        assert torch.equal(both[0], both[1])
        assert red.last_stats["bytes"] == 2 * sum(b - a for a, b in red.last_issued) and red.last_stats["collectives"] == len(red.last_issued)
    else:
        assert torch.allclose(arena, want, atol=1e-6), (rank, float((arena - want).abs().max()))
        assert red.last_stats["bytes"] == 4 * sum(b - a for a, b in red.last_issued) and red.last_stats["collectives"] == len(red.last_issued)
    dist.destroy_process_group()


@pytest.mark.parametrize("frozen_encoder", [False, True])
def test_bucketed_allreduce_mean_world2(frozen_encoder):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, frozen_encoder), nprocs=2, join=True)


def test_bucketed_allreduce_bf16_exchange_world2():
    """Optional bf16 gradient exchange (half the bytes per step): same stage / slice logic, the collective runs on a bf16
    image of each slice and the fp32 arena receives the mean -- within bf16 rounding of the fp32 exchange, identical on every rank."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, False, torch.bfloat16), nprocs=2, join=True)


def test_dasm_parameters_ride_the_decoder_and_heads_stages():
    """ddp.stage_of for the names transformer4sed_amd.dasm.DASM registers (detect_any_sound.py:149-260): a name that fell through to the last
    stage ("embed") would still be exchanged, but only after the whole backward -- the overlap the stage hooks exist for would be lost."""
    from transformer4sed_amd.ddp import stage_of
    for n in NAMES:
        if n.startswith(("at_decoder.", "at_projector.", "query_projector.", "at_query", "mask_embedding_layer.", "sed_head.", "at_head.")):
            assert stage_of(n, 12) == "decoder", n
    assert stage_of("norm_after_merge.weight", 12) == "heads"
    assert stage_of("at_query.1", 12) == "decoder"          # multi-modal query lists: one parameter per modality


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


class _P:
    def __init__(self, rg=True):
        self.requires_grad = rg


class _FakeNet2(_FakeNet):
    def __init__(self):
        self.params = {n: _P() for n in NAMES}

    def named_parameters(self):
        return list(self.params.items())


def test_small_stages_are_carried_and_merged(monkeypatch):
    """`min_bytes`: a stage smaller than the threshold does not get a collective of its own -- it rides with the next stage (adjacent
    slices merged) or with the end-of-backward flush; nothing is dropped and nothing is reduced twice."""
    from transformer4sed_amd.ddp import GradBucketReducer
    stages = ["decoder", "heads", ("block", 1), ("block", 0), "embed"]

    def run(min_bytes):
        net, opt = _FakeNet2(), _FakeOpt(_layout())
        red = GradBucketReducer(net, opt, min_bytes=min_bytes)
        red.force = True
        calls, at_hook = [], []
        monkeypatch.setattr(red, "_reduce", lambda t: calls.append((t.data_ptr(), t.numel())))
        net._last_grad_arena = torch.zeros(opt.total)
        base = net._last_grad_arena.data_ptr()
        for st in stages:
            red.on_stage(st)
            at_hook.append(len(calls))
        red.allreduce_grads()
        spans = sorted(((p - base) // 4, (p - base) // 4 + k) for p, k in calls)
        pos = 0
        for a, b in spans:          # a partition of the arena: complete, no overlap
            assert a == pos
            pos = b
        assert pos == opt.total
        assert red.carry == [] and red.fired == set()
        return at_hook, len(calls)
    eager, n_eager = run(0)
    # threshold just above the first stage's own size ("decoder": the context network, the heads and -- for DASM -- the query decoder)
    first = sum(b - a for a, b in GradBucketReducer(_FakeNet2(), _FakeOpt(_layout()), min_bytes=0).ranges["decoder"])
    lazy, n_lazy = run(4 * (first + 1))
    assert eager[0] > 0 and lazy[0] == 0          # the first (small) stage waits for company
    assert n_lazy < n_eager                        # carried slices merge with their arena neighbours


def test_ranges_follow_requires_grad_changes(monkeypatch):
    """Layer-wise unfreezing: a parameter un-frozen after the reducer was built must join the exchange."""
    from transformer4sed_amd.ddp import GradBucketReducer
    net, opt = _FakeNet2(), _FakeOpt(_layout())
    net.params["backbone.blocks.0.attn.qkv.weight"].requires_grad = False
    net.params["backbone.blocks.0.mlp.fc1.weight"].requires_grad = False
    red = GradBucketReducer(net, opt, min_bytes=0)
    assert ("block", 0) not in red.ranges
    net.params["backbone.blocks.0.mlp.fc1.weight"].requires_grad = True
    net._last_grad_arena = torch.zeros(opt.total)
    red.force = True
    calls = []
    monkeypatch.setattr(red, "_reduce", lambda t: calls.append(t.numel()))
    red.on_stage("decoder")
    assert ("block", 0) in red.ranges and len(red.ranges[("block", 0)]) == 1
    red.allreduce_grads()
    o, k = [(o, k) for n, o, k in opt.layout if n == "backbone.blocks.0.mlp.fc1.weight"][0]
    assert any(a <= o and o + k <= b for a, b in red.last_issued)


def test_inert_lr0_groups_stay_out_of_the_exchange():
    """Parameters of lr-0 groups (FusedAdamWEMA marks them inert: no gradient is computed for them) are not all-reduced, and joining /
    leaving that set rebuilds the ranges like a requires_grad change does."""
    from transformer4sed_amd.ddp import GradBucketReducer
    net, opt = _FakeNet2(), _FakeOpt(_layout())
    net._inert_param_names = frozenset(n for n in NAMES if n.startswith("backbone.blocks.0."))
    red = GradBucketReducer(net, opt, min_bytes=0)
    assert ("block", 0) not in red.ranges and ("block", 1) in red.ranges
    net._inert_param_names = frozenset()
    net._last_grad_arena = torch.zeros(opt.total)
    red.on_stage("decoder")
    assert ("block", 0) in red.ranges
    red.allreduce_grads()


def test_rank_sharded_batch_sampler():
    """Per-rank batch stream (SURVEY 8(e) 'Partitioning'): disjoint indices, the reference's strong | weak | unlabeled order on every
    rank, equal group sizes, and the union over ranks of batch i == the reference sampler's batch i."""
    from torch.utils.data import RandomSampler, SequentialSampler
    from transformer4sed_amd.data import ConcatDatasetBatchSampler, RankShardedBatchSampler
    sizes, bs, world = [40, 24, 64], [4, 4, 8], 2

    def samplers(seq):
        return [SequentialSampler(range(n)) if seq else RandomSampler(range(n)) for n in sizes]
    ref = list(ConcatDatasetBatchSampler(samplers(True), bs))
    ranks = [list(RankShardedBatchSampler(samplers(True), bs, rank=r, world=world)) for r in range(world)]
    assert len(ranks[0]) == len(ref) == 6
    offs = [0, 40, 64, 128]
    for i, gb in enumerate(ref):
        parts = [ranks[r][i] for r in range(world)]
        assert all(len(p) == sum(bs) // world for p in parts)
        assert not set(parts[0]) & set(parts[1])
        pos = 0
        for g, size in enumerate(bs):       # group by group: rank-major concatenation of the shares = the global group
            per = size // world
            share = [p[pos // world: pos // world + per] for p in parts]
            assert share[0] + share[1] == gb[pos:pos + size]
            assert all(offs[g] <= j < offs[g + 1] for sh in share for j in sh)
            pos += size
    # shuffled samplers: the same permutation on every rank (seeded per epoch), a different one next epoch
    r0 = RankShardedBatchSampler(samplers(False), bs, rank=0, world=world, seed=7)
    r1 = RankShardedBatchSampler(samplers(False), bs, rank=1, world=world, seed=7)
    e0 = [list(r0), list(r1)]
    seen = [j for b0, b1 in zip(*e0) for j in b0 + b1]
    assert len(seen) == len(set(seen))
    e1 = [list(r0), list(r1)]         # a second pass WITHOUT set_epoch (the reference loop never calls it) is a new epoch on every rank
    assert e1[0] != e0[0]
    seen = [j for b0, b1 in zip(*e1) for j in b0 + b1]
    assert len(seen) == len(set(seen))
    r0.set_epoch(0); r1.set_epoch(0)
    assert [list(r0), list(r1)] == e0
    with pytest.raises(ValueError):      # shuffling samplers without a shared seed: every rank would draw its own permutation
        RankShardedBatchSampler(samplers(False), bs, rank=0, world=world)
    with pytest.raises(ValueError):
        RankShardedBatchSampler(samplers(True), [3, 4, 8], rank=0, world=2)


def _bn_worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transformer4sed_amd.ddp import broadcast_buffers
    net = torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3), torch.nn.BatchNorm2d(4))
    torch.manual_seed(10 + rank)
    net.train()
    for _ in range(rank + 2):       # ranks see different data and even different step counts
        net(torch.randn(3, 1, 8, 8))
    mine = {k: v.clone() for k, v in net.named_buffers()}
    n = broadcast_buffers(net, src=0)
    assert n == 3
    got = torch.cat([b.reshape(-1).double() for _, b in net.named_buffers()])
    both = [torch.empty_like(got) for _ in range(world)]
    dist.all_gather(both, got)
    assert torch.equal(both[0], both[1])
    if rank == 0:       # rank 0 keeps its own statistics bit for bit, num_batches_tracked included
        assert all(torch.equal(mine[k], v) for k, v in net.named_buffers())
    else:
        assert int(net[1].num_batches_tracked) == 2 and net[1].num_batches_tracked.dtype == torch.int64
    dist.destroy_process_group()


def test_batchnorm_buffers_follow_rank0():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_bn_worker, args=(2, port), nprocs=2, join=True)


def test_bench_refuses_to_measure_fewer_ranks_than_asked(tmp_path):
    """`bench.py --gpus N` never reports a one-rank number as an N-GPU one: without enough HIP devices it stops before any work, and a
    launcher that started a different number of ranks than `--gpus` says is an error (missing item 1)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SED_BENCH_ONE_GPU")}
    env.update(CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and "HIP device" in (r.stderr + r.stdout) and "{" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=600)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout) and "{" not in r.stdout
