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
    # DASM (dasm.py): the query decoder / dual-stream head run their whole backward before the SED decoder's -- final at the "decoder" hook
    if name.startswith(("at_decoder.", "at_head.", "at_projector.", "at_query", "query_projector.", "mask_embedding_layer.", "sed_head.")):
        return "decoder"
    if name.startswith("at_adpater") or name.startswith("out_norm") or name.startswith("backbone.norm") or name.startswith("norm_after_merge"):
        return "heads"
    return "embed"


def broadcast_buffers(net, src=0, group=None):
    """BatchNorm running statistics (PaSST_CNN's CNN branch, src/models/cnn/base.py:75) under one process per GPU.

    POLICY: rank `src`'s buffers are the model's.  `nn.DataParallel` (recipes/desed/pmam/main.py:165) gives the same answer: every
    replica normalises its own slice with its own batch statistics, and only replica 0 -- which shares storage with the wrapped module
    -- keeps its running-statistics update, so the checkpointed / validated model carries the statistics of GPU 0's share of every
    batch.  Here each rank's buffers drift apart during training (batch statistics stay per rank, as in the reference); before anything
    reads them as "the model" (validation, checkpoint) every rank takes rank 0's, in ONE collective over a flat staging tensor."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    # only buffers that training mutates: the running statistics / batch counters of BatchNorm layers (constant tables -- the
    # frontend's window / twiddles -- are identical on every rank by construction and stay out of the collective)
    bufs = [b for mod in net.modules() if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm)
            for b in mod.buffers(recurse=False) if b is not None and b.numel() > 0]
    if not bufs:
        return 0
    flat = torch.cat([b.detach().reshape(-1).to(torch.float64) for b in bufs])   # float64 carries num_batches_tracked (int64) exactly
    if dist.get_backend(group) == "nccl":
        dist.broadcast(flat, src=src, group=group)
    else:
        host = flat.cpu()
        dist.broadcast(host, src=src, group=group)
        flat = host.to(flat.device)
    off = 0
    with torch.no_grad():
        for b in bufs:
            b.copy_(flat[off:off + b.numel()].view(b.shape).to(b.dtype))
            off += b.numel()
    if hasattr(net, "_param_generation"):
        net._param_generation += 1
    return len(bufs)


class GradBucketReducer:
    _budget_owner = None      # the reducer whose CU reservation is in force (process-global setting)

    def __init__(self, net, optimizer, group=None, min_bytes=8 << 20, comm_dtype=None, reserve_cus=None):
        """`comm_dtype`: torch.float32 (default) exchanges the fp32 arena slices in place; torch.bfloat16 exchanges a bf16 image of every
        slice -- half the bytes on the xGMI links (200 MB instead of 400 MB per finetune2 step) for one rounding of the
        summed gradient to 8 significand bits, the precision its MFMA operands had anyway.  The image is cast on the compute stream when
        the stage fires, reduced on the collective stream, and written back into the fp32 arena before the optimiser reads it.
        Environment default: SED_DDP_COMM_DTYPE=bf16|fp32.
        `reserve_cus`: CUs left to the communication kernels while this reducer lives -- until `close()` / the end of a `with` block / its
        destruction restores the full grid (sed_gemm_set_cu_budget(total - reserve) now, (0) then; default 0).  With the dynamic tile walk of the persistent GEMMs a reserve is not needed for correctness of the
        overlap -- a late workgroup costs nothing (profiles/r4_cu_steal.txt) -- it only avoids queueing workgroups that will find no tile."""
        import os
        self.net, self.opt, self.group = net, optimizer, group
        if comm_dtype is None:
            name = os.environ.get("SED_DDP_COMM_DTYPE", "fp32")
            if name not in ("bf16", "fp32"):
                raise ValueError(f"SED_DDP_COMM_DTYPE={name!r}: expected bf16 or fp32")
            comm_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32}[name]
        if comm_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("comm_dtype must be torch.float32 or torch.bfloat16")
        self.comm_dtype = comm_dtype
        self._stage = None            # bf16 staging arena (same offsets as the gradient arena), allocated on first use
        self.stats = dict(collectives=0, bytes=0)        # of the current backward; `last_stats` = of the last finished one
        self.last_stats = dict(collectives=0, bytes=0)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if reserve_cus is None:
            reserve_cus = 0
        self.reserve_cus = int(reserve_cus)
        self._budget_set = False
        if self.reserve_cus > 0 and self.world > 1 and torch.cuda.is_available():
            from .ops import call
            total = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
            call("sed_gemm_set_cu_budget", max(8, total - self.reserve_cus))
            self._budget_set = True
            GradBucketReducer._budget_owner = self
        # a stage whose slices are smaller than this is not worth a collective of its own: it is carried to the next stage hook (or
        # to the end of backward) and merged with adjacent slices there
        self.min_elems = min_bytes // 4
        self.ranges = {}
        self._flags = None
        self._build_ranges()
        self.pending = []
        self.carry = []
        self.fired = set()
        self.order = []       # stages in the order their hooks fired during the current backward
        self.issued = []      # [start, end) arena ranges of the collectives of the current backward, in issue order
        self.force = False  # issue the collectives even at world size 1 (single-GPU check of the RCCL path)
        net._grad_ready_hook = self.on_stage
        self.use_avg = dist.is_initialized() and dist.get_backend(group) == "nccl"

    def _trainable_flags(self):
        # (parameters of lr-0 groups are inert: the model computes no gradient for them, see FusedAdamWEMA.skip_zero_lr_groups)
        inert = getattr(self.net, "_inert_param_names", ())
        return tuple(p.requires_grad and n not in inert for n, p in self.net.named_parameters()) if hasattr(self.net, "named_parameters") else None

    def _build_ranges(self):
        """Arena slices per backward stage.  Only slices that can receive a gradient take part: frozen parameters and PaSST's unused
        classification heads (`backbone.head*`: never on the MAT-SED path, their arena slices stay zero) are excluded.  Rebuilt when
        the set of trainable parameters changes (layer-wise unfreezing), see `_refresh`."""
        net = self.net
        self._flags = self._trainable_flags()
        inert = getattr(net, "_inert_param_names", ())
        trainable = {n for n, p in net.named_parameters() if p.requires_grad and n not in inert} if hasattr(net, "named_parameters") else None
        self.ranges = {}
        for n, o, k in self.opt.layout:
            if n.startswith("backbone.head") or (trainable is not None and n not in trainable):
                continue
            st = stage_of(n, getattr(net, "depth", 12))
            self.ranges.setdefault(st, []).append([o, o + (k + 63) // 64 * 64])
        for st, rs in self.ranges.items():
            self.ranges[st] = self._merge(rs)

    @staticmethod
    def _merge(rs):
        rs = sorted(rs)
        merged = [list(rs[0])]
        for a, b in rs[1:]:
            if a == merged[-1][1]:
                merged[-1][1] = b
            else:
                merged.append([a, b])
        return merged

    def _refresh(self):
        """A parameter un-frozen (or frozen) since the ranges were built would otherwise silently drop out of (or stay in) the
        exchange and let the ranks diverge: compare the requires_grad flags (220 booleans) once per backward."""
        if not self.fired and not self.carry and self._trainable_flags() != self._flags:
            self._build_ranges()

    def _reduce(self, t, back=None):
        """all-reduce (mean) of `t` in place; `back` = (fp32 arena slice) when `t` is its bf16 image: written back after the wait."""
        self.stats["collectives"] += 1
        self.stats["bytes"] += t.numel() * t.element_size()
        if self.world == 1 and not self.force:
            if back is not None:
                back.copy_(t)
            return
        if self.use_avg:
            self.pending.append((dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None, t, back))
        else:
            w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.pending.append((w, self.world, t, back))

    def _issue(self, ranges):
        arena = self.net._last_grad_arena
        for a, b in self._merge(ranges):
            self.issued.append((a, b))
            if self.comm_dtype == torch.float32 or (self.world == 1 and not self.force):
                self._reduce(arena[a:b])      # (a single rank exchanges nothing: no bf16 round trip of its own gradient either)
            else:
                if self._stage is None or self._stage.numel() != arena.numel() or self._stage.device != arena.device:
                    self._stage = torch.empty(arena.numel(), dtype=self.comm_dtype, device=arena.device)
                img = self._stage[a:b]
                img.copy_(arena[a:b])                 # cast on the compute stream, behind the kernels that produced the slice
                self._reduce(img, back=arena[a:b])

    def on_stage(self, stage):
        """Called by the engine as soon as every gradient of `stage` is final."""
        self._refresh()
        self.fired.add(stage)
        self.order.append(stage)
        todo = self.carry + [list(r) for r in self.ranges.get(stage, [])]
        if sum(b - a for a, b in todo) < self.min_elems:
            self.carry = todo
            return
        self.carry = []
        if todo:
            self._issue(todo)

    def wait_pending(self):
        """Block the compute stream on every collective issued so far (the averaged slices are final afterwards)."""
        for w, div, t, back in self.pending:
            w.wait()
            if div is not None:
                t.div_(div)
            if back is not None:
                back.copy_(t)                         # bf16 image -> fp32 arena (what the optimiser reads)
        self.pending = []

    def allreduce_grads(self, net=None):
        """After backward: reduce whatever no stage hook covered (frozen stages never fire) and what was carried, then wait."""
        self._refresh()
        todo = self.carry
        self.carry = []
        for st, rs in self.ranges.items():
            if st not in self.fired:
                todo = todo + [list(r) for r in rs]
        if todo:
            self._issue(todo)
        self.wait_pending()
        self.fired = set()
        self.order = []
        self.last_issued, self.issued = self.issued, []
        self.last_stats, self.stats = self.stats, dict(collectives=0, bytes=0)

    def close(self):
        """Give the GEMMs their full grid back (the CU budget is process-global: a reducer that reserved CUs for the collectives must not
        leave validation-only or single-GPU code of the same process on the reduced grid) and detach from the model."""
        if self._budget_set and GradBucketReducer._budget_owner is self:      # (only the reducer that made the current reservation undoes it)
            from .ops import call
            call("sed_gemm_set_cu_budget", 0)
            GradBucketReducer._budget_owner = None
        self._budget_set = False
        if getattr(self.net, "_grad_ready_hook", None) == self.on_stage:
            self.net._grad_ready_hook = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        # No launch from a finaliser: the CU budget is process-global, and a reducer collected after its successor was constructed would wipe
        # the successor's reservation (or call into the HIP library during interpreter shutdown).  `close()` / the `with` block are the way
        # to give the CUs back; a reducer that is simply dropped only detaches its hook.
        try:
            if getattr(self.net, "_grad_ready_hook", None) == self.on_stage:
                self.net._grad_ready_hook = None
        except Exception:
            pass

    def sync_buffers(self, src=0):
        """Rank `src`'s BatchNorm running statistics become every rank's (see `broadcast_buffers` for the policy)."""
        return broadcast_buffers(self.net, src=src, group=self.group)
