"""Data-parallel QAT: one process per GPU, gradients all-reduced over RCCL (xGMI) -- replaces the reference's
single-process ``nn.DataParallel`` (``*/main.py``: scatter / broadcast of all parameters every step / gather /
reduce-add to GPU 0).

Design for the MI355X node (7 xGMI links per GPU, point-to-point): parameters live on every rank, so the only
exchange per step is the gradient all-reduce.  Gradients are packed into a few large flat buckets in reverse
parameter order (the order backward produces them); a bucket's all-reduce is launched asynchronously the moment its
last gradient is accumulated, overlapping the rest of backward; ``wait()`` before ``optimizer.step()`` re-points each
``p.grad`` at its slice of the reduced bucket (no copy back).  nin_gc (2.37 MB of gradients) is one latency-bound
collective; resnet18 (44.7 MB) splits into two.

Cross-rank statistics: the parity target of data-parallel QAT is the single-process reference on the concatenated global batch.
  * gradients: mean all-reduce == full-batch gradient (mean-reduced loss, equal shards);
  * IAO activation / ``QuantAdd`` observers (level 'L'): ``sync_observers(model)`` makes every such observer all-reduce the CURRENT batch's
    (min, max) over the ranks (one 2-float MAX collective on [-min, max]) BEFORE its running-extreme / moving-average update and the
    ``update_qparams`` behind it (wqaq/iao/quantize.py:214-240, 1484-1498), so every rank quantises with the global-batch range -- exactly what the
    single process sees, since min / max over a batch decompose over shards.  One tiny collective per activation quantizer per forward: each range
    is needed before the next layer can run, so they cannot be packed;
  * BatchNorm / BN-fuse batch mean and variance: per-rank (what ``nn.BatchNorm2d`` does under DP / DDP);
  * weight observers and the DoReFa / wbwtab weight quantizers: rank-invariant (same weights everywhere).
"""
import os

import torch
import torch.distributed as dist


def active(group=None):
    """Is the data-parallel machinery (gradient all-reduce, observer range collectives) on: a process group with more than one rank -- or with ONE rank under
    MN_DP_SINGLE=1, i.e. the step one GPU of an N-GPU job executes with every collective issued and nothing on the links.  That is how bench.py puts a number on
    the data-parallel step's own cost (packing, collective launches, graph segments) on a one-GPU box."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get("MN_DP_SINGLE") == "1")


class _Bucket:
    __slots__ = ("flat", "params", "offsets", "pending", "handle")


class GradSync:
    def __init__(self, model, bucket_bytes=32 << 20, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        params = [p for p in model.parameters() if p.requires_grad]
        self.buckets, self._where, self._hooks = [], {}, []
        cur, size = [], 0
        for p in reversed(params):
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self._close(cur)
                cur, size = [], 0
        if cur:
            self._close(cur)
        self.on = active(group)
        if self.on:
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _close(self, params):
        b = _Bucket()
        b.params, b.offsets, n = list(params), [], 0
        for p in params:
            b.offsets.append(n)
            n += p.numel()
        b.flat = torch.empty(n, dtype=params[0].dtype, device=params[0].device)
        b.pending, b.handle = len(params), None
        for p, off in zip(params, b.offsets):
            self._where[id(p)] = (b, off)
        self.buckets.append(b)

    def _on_grad(self, p):
        b, off = self._where[id(p)]
        b.flat[off:off + p.numel()].copy_(p.grad.reshape(-1))
        b.pending -= 1
        if b.pending == 0:
            b.flat.div_(self.world)                       # mean over ranks == gradient of the mean loss over the global batch
            b.handle = dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self):
        """Call between ``loss.backward()`` and ``optimizer.step()``."""
        if not self.on:
            return
        for b in self.buckets:
            if b.pending != 0:
                raise RuntimeError("GradSync: a bucket did not receive all its gradients (unused parameter?)")
            b.handle.wait()
            for p, off in zip(b.params, b.offsets):
                p.grad = b.flat[off:off + p.numel()].view_as(p)
            b.pending, b.handle = len(b.params), None

    def remove(self):
        for h in self._hooks:
            h.remove()


def broadcast_parameters(model, src=0, group=None):
    """One-time sync at start-up (the reference re-broadcasts every forward)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)


class ReplicaBuffers:
    """The reference's own data-parallel semantics for module BUFFERS -- IAO observer ranges, scales and zero points, BatchNorm running statistics
    (wqaq/iao/main.py:496-500: ``nn.DataParallel`` re-creates the replicas from the module on device 0 in every forward, so each replica updates and uses its buffers from
    ITS shard of the batch and only device 0's survive the step): no collective inside forward; once per step, next to the gradient all-reduce, rank 0's floating-point
    buffers are broadcast as ONE flat tensor.  The alternative to ``sync_observers`` (every activation range reduced over the global batch before it is used: bit-identical
    to one process on the concatenated batch, one blocking collective per quantizer -- 30 per resnet18 step)."""

    def __init__(self, model, group=None, src=0):
        self.group, self.src = group, src
        self.bufs = [b for b in model.buffers() if b.is_floating_point() and b.numel() > 0]
        self.flat = None
        if self.bufs:
            dev = self.bufs[0].device
            self.flat = torch.empty(sum(b.numel() for b in self.bufs), dtype=torch.float32, device=dev)
            self.views, o = [], 0
            for b in self.bufs:
                self.views.append(self.flat[o:o + b.numel()].view(b.shape))
                o += b.numel()
            self._all_f32 = all(b.dtype == torch.float32 for b in self.bufs)
        model._mn_replica_buffers = self

    def exchange(self):
        """pack -> broadcast from rank 0 -> unpack (three launches and one small collective; a no-op without a process group)."""
        if self.flat is None or not active(self.group):
            return
        if self._all_f32:
            torch.cat([b.detach().reshape(-1) for b in self.bufs], out=self.flat)
        else:
            torch.cat([b.detach().to(torch.float32).reshape(-1) for b in self.bufs], out=self.flat)
        dist.broadcast(self.flat, src=self.src, group=self.group)
        with torch.no_grad():
            torch._foreach_copy_([b.detach() for b in self.bufs], self.views)


def replica_buffers(model, group=None):
    """Switch ``model`` to the reference's DataParallel buffer semantics (``ReplicaBuffers``) and make sure no observer reduces its range over the ranks."""
    sync_observers(model, group, enable=False)
    return ReplicaBuffers(model, group)


def train_step_dp(model, optimizer, sync, data, target):
    """``train_step`` with the gradient exchange between backward and the optimizer step."""
    import torch.nn.functional as F
    output = model(data)
    from micronet_amd.train import cross_entropy
    loss = cross_entropy(output, target)
    optimizer.zero_grad()
    loss.backward()
    sync.wait()
    rb = getattr(model, "_mn_replica_buffers", None)
    if rb is not None:
        rb.exchange()
    optimizer.step()
    return loss, output


# Set by train.GraphedTrainStep while it captures a model whose forward holds range collectives: a callable (buf, group) that ends the HIP graph being
# captured, notes ``buf`` as the operand of the collective to run after that segment's replay, and begins the next segment.
_segment_cut = None


def allreduce_minmax(min_t, max_t, group=None):
    """In place: min_t <- min over ranks, max_t <- max over ranks (tensors of equal shape, any device the backend supports) with ONE collective:
    MAX over the stacked [-min, max]."""
    if not active(group):
        return
    buf = torch.stack([-min_t.reshape(-1), max_t.reshape(-1)])
    if _segment_cut is not None:
        _segment_cut(buf, group)          # a step being captured in HIP-graph segments (train.GraphedTrainStep): the collective runs BETWEEN two replays
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=group)
    min_t.copy_((-buf[0]).view_as(min_t))
    max_t.copy_(buf[1].view_as(max_t))


def allreduce_range(buf, group=None):
    """In place on ``buf`` = [min_0, max_0, min_1, max_1, ...] (floats on the device): every min over the ranks, every max over the ranks, ONE MAX collective on
    [-min, max, ...].  What a synced IAO observer calls between its local reduction and its running update; inside a segmented graph capture the collective
    becomes a cut (``_segment_cut``)."""
    if not active(group):
        return
    buf[0::2].neg_()
    if _segment_cut is not None:
        _segment_cut(buf, group)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=group)
    buf[0::2].neg_()


def sync_observers(model, group=None, enable=True):
    """Switch the cross-rank range reduction of the ACTIVATION observers of ``model`` on: the observer of every IAO quantizer with ``activation_weight_flag == 1``
    and the two input observers of every ``QuantAdd`` (each sees only its rank's shard of the batch).  Weight observers are left alone -- the weights are the
    same on every rank, their ranges are rank-invariant and need no collective.  Returns how many observers were switched; a no-op for models without such
    observers (DoReFa, wbwtab)."""
    seen, n = set(), 0

    def switch(obs):
        nonlocal n
        if obs is None or id(obs) in seen or type(obs).__name__ == "HistogramObserver":
            return
        if getattr(obs, "q_level", None) == "L" and hasattr(obs, "min_val") and hasattr(obs, "max_val"):
            seen.add(id(obs))
            obs._mn_sync_group = group if enable else None
            obs._mn_sync = bool(enable)
            n += 1
    for m in model.modules():
        if getattr(m, "activation_weight_flag", None) == 1 and hasattr(m, "observer"):
            switch(m.observer)
        if hasattr(m, "observer_res") and hasattr(m, "observer_shortcut"):
            switch(m.observer_res)
            switch(m.observer_shortcut)
    return n
