"""Data-parallel gradient exchange for MAT-SED: one process per GPU, RCCL all-reduce over xGMI.

The reference uses single-process `nn.DataParallel` (recipes/desed/finetune/passt/main.py:31-33: parameters re-broadcast
every forward, gradients reduced to GPU0, frontend/loss/optimiser serialised on GPU0).  Here every rank runs the whole step
on its own clips; the only exchange is the mean of the gradients.  Because the model's backward writes all gradients into
ONE flat fp32 arena laid out like the optimiser's parameter arena (trainer.FusedAdamWEMA), buckets are contiguous slices:
the engine signals when a stage of the backward is final ("decoder", "heads", ("block", i), "embed") and the slices that
became complete are all-reduced asynchronously on the process group's stream while the rest of the backward keeps running
on the compute stream.  8 MI355X are fully connected (7 xGMI links/GPU): 400 MB of fp32 gradients per step is ~5 ms even
link-bound on a ring, far below the ~100 ms backward, so a handful of large slices is enough.
"""
import re

import torch
import torch.distributed as dist


def stage_of(name, depth):
    if name.startswith("backbone.blocks."):
        return ("block", int(re.match(r"backbone\.blocks\.(\d+)\.", name).group(1)))
    if name.startswith("decoder.") or name.startswith("classifier.") or name.startswith("mlm_mlp") or name == "mask_token":
        return "decoder"
    if name.startswith("at_adpater") or name.startswith("out_norm") or name.startswith("backbone.norm"):
        return "heads"
    return "embed"


class GradBucketReducer:
    def __init__(self, net, optimizer, group=None, min_bytes=8 << 20):
        self.net, self.opt, self.group = net, optimizer, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.min_elems = min_bytes // 4
        self.ranges = {}
        # Only slices that can receive a gradient take part: frozen parameters and PaSST's unused classification heads
        # (`backbone.head*`: never on the MAT-SED path, their arena slices stay zero) are excluded statically.
        trainable = {n for n, p in net.named_parameters() if p.requires_grad} if hasattr(net, "named_parameters") else None
        for n, o, k in optimizer.layout:
            if n.startswith("backbone.head") or (trainable is not None and n not in trainable):
                continue
            st = stage_of(n, getattr(net, "depth", 12))
            self.ranges.setdefault(st, []).append([o, o + (k + 63) // 64 * 64])
        for st, rs in self.ranges.items():  # merge adjacent slices
            rs.sort()
            merged = [rs[0]]
            for a, b in rs[1:]:
                if a == merged[-1][1]:
                    merged[-1][1] = b
                else:
                    merged.append([a, b])
            self.ranges[st] = merged
        self.pending = []
        self.fired = set()
        self.order = []       # stages in the order their hooks fired during the current backward
        self.force = False  # issue the collectives even at world size 1 (single-GPU check of the RCCL path)
        net._grad_ready_hook = self.on_stage
        self.use_avg = dist.is_initialized() and dist.get_backend(group) == "nccl"

    def _reduce(self, t):
        if self.world == 1 and not self.force:
            return
        if self.use_avg:
            self.pending.append(dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
        else:
            w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.pending.append((w, t))

    def on_stage(self, stage):
        """Called by the engine as soon as every gradient of `stage` is final."""
        arena = self.net._last_grad_arena
        self.fired.add(stage)
        self.order.append(stage)
        for a, b in self.ranges.get(stage, []):
            self._reduce(arena[a:b])

    def wait_pending(self):
        """Block the compute stream on every collective issued so far (the averaged slices are final afterwards)."""
        for w in self.pending:
            if isinstance(w, tuple):
                w[0].wait()
                w[1].div_(self.world)
            else:
                w.wait()
        self.pending = []

    def allreduce_grads(self, net=None):
        """After backward: reduce whatever no stage hook covered (frozen stages never fire), then wait."""
        arena = self.net._last_grad_arena
        for st, rs in self.ranges.items():
            if st not in self.fired:
                for a, b in rs:
                    self._reduce(arena[a:b])
        self.wait_pending()
        self.fired = set()
        self.order = []
