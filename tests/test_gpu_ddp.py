"""Real-model data-parallel checks on ONE GPU (SURVEY 8(e) "Validation"; replaces nn.DataParallel of
recipes/desed/finetune/passt/main.py:31-33 and the EMA / optimiser step of src/utils/__init__.py:11-21, train.py:197-201).

Two ranks share cuda:0 over gloo (the exchange is the same `GradBucketReducer` code path as RCCL; only the transport differs):
  (a) after the stage-triggered all-reduce every rank's gradient arena equals the single-process gradient on the concatenated
      batch (mean-reduced losses => mean of the per-rank gradients),
  (b) every stage hook ("decoder", "heads", ("block", i) top-down, "embed") fired, in backward order, before `allreduce_grads`,
  (c) after the fused AdamW + EMA step the parameter and EMA arenas are bit-identical across ranks.
A world-size-1 NCCL (RCCL) run checks that the collectives are issued on the process group's own stream: a collective issued
before a long compute kernel completes while that kernel is still running."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

DEPTH = 2
TOL = 5e-6    # measured 1.5e-6
CFG = {"encoder": {"lr": 5e-6, "weight_decay": 1e-4, "freeze_layer": 0, "step_lr": 4},
       "decoder": {"lr": 1e-4, "weight_decay": 1e-4}, "head": {"lr": 1e-4, "weight_decay": 1e-4}}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build(dev):
    import copy
    from transformer4sed_amd import synth
    from transformer4sed_amd.passt_sed import PaSST_SED
    from transformer4sed_amd.trainer import FusedAdamWEMA, get_params
    net = PaSST_SED(passt_feature_layer=DEPTH, f_pool="mean_pool", decode_ratio=10, at_adapter=True, decoder="transformerXL",
                    decoder_layer_num=3, decoder_pos_emd_len=1000, mlm=False, load_pretrained_model=False, encoder_depth=DEPTH)
    sd = synth.matsed_state_dict_np(tag="w768", depth=12)
    own = net.state_dict()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items() if k in own}, strict=True)
    net = net.to(dev).train()
    ema = copy.deepcopy(net)
    for p in ema.parameters():
        p.detach_()
    opt = FusedAdamWEMA(net, get_params(net, CFG), ema_net=ema)
    return net, ema, opt


def _inputs(dev, n):
    g = torch.Generator().manual_seed(11)
    mel = (torch.randn(n, 128, 1000, generator=g) * 2.0).to(dev)
    strong = (torch.rand(n, 10, 1000, generator=g) < 0.2).float().to(dev)
    weak = (strong.sum(-1) > 0).float()
    return mel, strong, weak


def _loss(net, mel, strong, weak):
    s, w, other = net(mel)
    bce = torch.nn.functional.binary_cross_entropy
    return bce(s, strong) + 0.5 * bce(w, weak) + 0.5 * bce(other["at_out"], weak) + 0.1 * (s * s).mean()


def _worker(rank, world, port, per_rank, q, comm=None):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from transformer4sed_amd.ddp import GradBucketReducer
        net, ema, opt = _build(dev)
        red = GradBucketReducer(net, opt, comm_dtype=comm)
        # static exclusions: PaSST's unused heads are not part of any bucket
        covered = sum(b - a for rs in red.ranges.values() for a, b in rs)
        dead = sum((k + 63) // 64 * 64 for n, o, k in opt.layout if n.startswith("backbone.head"))
        assert dead > 0 and covered == opt.total - dead, (covered, opt.total, dead)
        mel, strong, weak = _inputs(dev, per_rank * world)
        lo, hi = rank * per_rank, (rank + 1) * per_rank
        # ---- single-process reference gradient on the concatenated batch (no reducer attached)
        net._grad_ready_hook = None
        _loss(net, mel, strong, weak).backward()
        ref = net._last_grad_arena.clone()
        opt.zero_grad()
        # ---- this rank's slice, through the reducer
        net._grad_ready_hook = red.on_stage
        seen = []
        orig = red.on_stage

        def spy(stage):
            seen.append(stage)
            orig(stage)
        net._grad_ready_hook = spy
        _loss(net, mel[lo:hi], strong[lo:hi], weak[lo:hi]).backward()
        # (b) all stages fired during the backward, top-down
        want = ["decoder", "heads"] + [("block", i) for i in range(DEPTH - 1, -1, -1)] + ["embed"]
        assert seen == want, seen
        assert len(red.pending) > 0
        red.allreduce_grads(net)
        got = net._last_grad_arena
        # (a) mean of the per-rank gradients == gradient of the batch-mean loss on all clips
        errs = []
        for n, o, k in opt.layout:
            a, b = got[o:o + k], ref[o:o + k]
            scale = float(b.abs().max())
            if scale == 0.0:
                assert float(a.abs().max()) == 0.0, n
                continue
            errs.append((float((a - b).abs().max()) / scale, n))
        errs.sort(reverse=True)
        if rank == 0:
            print("largest relative gradient differences (2 ranks vs single process):", errs[:6], flush=True)
        # Per-clip forward / dX are bit-identical for any batch composition; the token sums of the weight gradients are
        # re-associated (split-K partition depends on the token count): a few fp32 ulps of the largest element.  One tensor is not
        # linear in the batch at that level: d(linear_pos.weight) is a GEMM over dP, the rel-pos gradient summed over the clips
        # BEFORE it is rounded to a bf16 MFMA operand, so round(sum over 4 clips) != mean of round(sum over 2 clips) (bf16 ulp 4e-3).
        err = max(e for e, n in errs if "linear_pos" not in n)
        if comm is None and red.comm_dtype == torch.bfloat16:
            comm = torch.bfloat16           # SED_DDP_COMM_DTYPE=bf16 in the environment (tools/nondefault_suite.sh): the reducer's default
        if comm == torch.bfloat16:      # the exchanged images are bf16: one rounding per rank and one of the mean (2^-9 relative each)
            assert err < 2 ** -7, errs[:4]
            assert red.last_stats["bytes"] == 2 * covered, (red.last_stats, covered)
        else:
            assert err < TOL, errs[:4]
            assert red.last_stats["bytes"] == 4 * covered, (red.last_stats, covered)
        assert max(e for e, n in errs if "linear_pos" in n) < (2 ** -7 if comm == torch.bfloat16 else 4e-3), errs[:4]
        # (c) step, then compare parameters and EMA across ranks bit for bit
        opt.step(0.99)
        torch.cuda.synchronize()
        for arena in (opt.arena, opt.ema_arena):
            mine = arena.detach().cpu()
            both = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            assert torch.equal(both[0], both[1])
        assert not torch.equal(opt.arena.cpu(), opt.ema_arena.cpu())
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", err))
    except Exception as e:  # surface the failure in the parent
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
        raise


@pytest.mark.parametrize("comm", [None, torch.bfloat16], ids=["fp32-exchange", "bf16-exchange"])
def test_two_rank_gradients_match_single_process_and_replicas_stay_identical(comm):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 2, q, comm)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    res = [q.get(timeout=10) for _ in procs]
    assert all(r[1] == "ok" for r in res), res


def test_rccl_collectives_run_beside_the_compute_stream():
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from transformer4sed_amd.ddp import GradBucketReducer
        net, ema, opt = _build(dev)
        red = GradBucketReducer(net, opt)
        red.force = True                     # issue the collectives at world size 1
        mel, strong, weak = _inputs(dev, 2)
        _loss(net, mel, strong, weak).backward()
        single = net._last_grad_arena.clone()
        assert len(red.pending) >= DEPTH + 3    # one or more slices per stage, all in flight behind the backward
        red.allreduce_grads(net)
        torch.cuda.synchronize()
        if red.comm_dtype == torch.bfloat16:      # (SED_DDP_COMM_DTYPE=bf16: the forced exchange goes through the bf16 image)
            scale = float(single.abs().max())
            assert float((net._last_grad_arena - single).abs().max()) <= 2 ** -8 * scale
        else:
            assert torch.equal(net._last_grad_arena, single)      # AVG over one rank is the identity
        # stream check: a collective issued BEFORE a long compute kernel finishes while that kernel is still running
        big = torch.randn(8192, 8192, device=dev, dtype=torch.float16)
        buf = torch.ones(1 << 20, device=dev)
        torch.cuda.synchronize()
        done = torch.cuda.Event()
        work = dist.all_reduce(buf, op=dist.ReduceOp.AVG, async_op=True)
        for _ in range(60):                   # ~50 ms of matmuls on the compute stream
            big @ big
        done.record()
        import time
        t0 = time.time()
        while not work.is_completed() and time.time() - t0 < 5.0:
            time.sleep(0.0005)
        assert work.is_completed()
        assert not done.query(), "the collective only completed after the compute stream drained: it is not on its own stream"
        work.wait()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ whole trainers under two ranks
def _gather_equal(t, world):
    import torch.distributed as dist
    mine = t.detach().cpu()
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    return all(torch.equal(both[0], b) for b in both[1:])


def _trainer_worker(rank, world, port, kind, q):
    """Two ranks on cuda:0 (gloo): the REAL train step -- frontend, augmentation with rank-specific draws, student forward / backward
    with stage hooks + side-stream weight gradients, teacher windows on the second stream, reducer, fused AdamW + EMA -- twice; then
    parameters, EMA and Adam moments must be bit-identical on both ranks although every rank saw different clips."""
    try:
        import json
        import random
        import sys
        import numpy as np
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        from transformer4sed_amd import synth
        from transformer4sed_amd.ddp import GradBucketReducer
        random.seed(50 + rank); np.random.seed(50 + rank); torch.manual_seed(50 + rank)
        if kind == "matsed":
            B = 6
            net, ema_net, opt, trainer, _ = bench.build(B, DEPTH, dev, "finetune2")
            trainer.cfg = json.loads(json.dumps(bench.FINETUNE2))
            trainer.cfg["training"]["batch_size"] = [2, 0, 2, 2]
            labels = torch.from_numpy(synth.synth_batch_labels(2, 2, 2, seed=60 + rank)).to(dev)
            step = lambda wav: trainer.finetune_step(wav, labels.clone())
            # the teacher's windows run on the second stream (and the dW GEMMs on the side stream) -- unless the single-stream profiling
            # switch of tools/nondefault_suite.sh turned that off for this run
            assert trainer.overlap_teacher or os.environ.get("SED_OVERLAP_TEACHER") == "0"
        elif kind == "dasm":      # DASMTrainer.train (row (g)): query decoder + dual-stream head gradients ride the "decoder" stage
            B = 3
            net, opt, trainer = bench.build_dasm_train(DEPTH, dev, 12)
            ema_net = None
            labels = torch.from_numpy(synth.synth_strong_labels(B, n_classes=12, seed=60 + rank)).to(dev)
            step = lambda wav: trainer.step(wav, labels.clone())
        else:
            B = 4
            net, opt, trainer = bench.build_pmam(DEPTH, dev)
            ema_net = None
            labels = torch.from_numpy(synth.synth_strong_labels(B, n_classes=30, seed=60 + rank)).to(dev)
            step = lambda wav: trainer.step(wav, labels.clone())
        trainer.ddp = GradBucketReducer(net, opt)
        start = opt.arena.detach().clone()
        assert _gather_equal(opt.arena, world), "replicas must start from the same weights"
        losses = []
        for it in range(2):
            wav = torch.from_numpy(synth.synth_wav(B, seed=70 + 10 * it + rank)).to(dev)
            out = step(wav)
            losses.append(float(out["loss_total"]))
            assert trainer.ddp.last_issued, "no gradient slice was exchanged"
        torch.cuda.synchronize()
        assert all(np.isfinite(losses))
        both = [None] * world
        dist.all_gather_object(both, losses)
        assert both[0] != both[1], "the ranks were meant to see different clips"
        assert not torch.equal(start, opt.arena)
        for name, t in (("parameters", opt.arena), ("adam m", opt.m), ("adam v", opt.v)) + ((("EMA", opt.ema_arena),) if ema_net is not None else ()):
            assert _gather_equal(t, world), f"{name} differ between the ranks after two steps"
        if kind == "pmam":
            # BatchNorm running statistics are per rank while training ...
            rm = dict(net.named_buffers())["cnn.cnn.batchnorm0.running_mean"]
            assert not _gather_equal(rm, world)
            # ... and rank 0's become the model's as soon as something reads the model (validation / checkpoint)
            mine0 = rm.detach().clone()
            pm = torch.zeros(B, 1000, dtype=torch.bool, device=dev)
            trainer.cfg.setdefault("PaSST_CNN", {}).setdefault("val_kwargs", {"encoder_win": False, "temp_w": 1})
            trainer.validation_step(torch.from_numpy(synth.synth_wav(B, seed=99)).to(dev), labels.clone(), pm)
            for n, b in net.named_buffers():
                assert _gather_equal(b, world), n
            if rank == 0:
                assert torch.equal(mine0, rm)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok", losses))
    except Exception:
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
        raise


@pytest.mark.parametrize("kind", ["matsed", "pmam", "dasm"])
def test_two_rank_trainer_steps_keep_replicas_identical(kind):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, kind, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(900)
    res = [q.get(timeout=10) for _ in procs]
    assert all(r[1] == "ok" for r in res), res


def test_bench_gpus2_starts_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it becomes two ranks (here both on cuda:0 over gloo -- RCCL refuses two ranks
    per device) and reports the LIVE world size; asking for more GPUs than the box has fails loudly instead of measuring one."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    args = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--depth", "2", "--batch", "6", "--steps", "2", "--warmup", "1",
            "--no-cpu-baseline"]
    out = subprocess.run(args, capture_output=True, text=True, timeout=900, cwd=root,
                         env=dict(env, SED_BENCH_BACKEND="gloo", SED_BENCH_ONE_GPU="1"))
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["ranks"] == 2 and line["config"]["global_batch"] == 12 and line["config"]["parallelism"] == "dp2"
    assert line["value"] > 0 and abs(line["value"] - 12 * 2 / (line["ms_per_step"] * 2 / 1000)) < 1e-2 * line["value"]
    if torch.cuda.device_count() < 2:
        out = subprocess.run(args, capture_output=True, text=True, timeout=300, cwd=root, env=env)
        assert out.returncode != 0 and "HIP device" in (out.stderr + out.stdout)
