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


def _worker(rank, world, port, per_rank, q):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from transformer4sed_amd.ddp import GradBucketReducer
        net, ema, opt = _build(dev)
        red = GradBucketReducer(net, opt)
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
        assert err < TOL, errs[:4]
        assert max(e for e, n in errs if "linear_pos" in n) < 4e-3, errs[:4]
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


def test_two_rank_gradients_match_single_process_and_replicas_stay_identical():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 2, q)) for r in range(2)]
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
