"""Training-step harness: the caller of the hot path, mirrored from the reference's ``main.py`` scripts.

``build_model`` + ``init_like_main`` reproduce ``wqaq/dorefa/main.py:200,283-297`` (seed 1, xavier conv weights,
zero biases, N(0, 0.01) linear weights); ``make_optimizer`` is ``main.py:308-315`` (Adam, one param group per
tensor); ``train_step`` is the loop body ``main.py:77-82``; ``synth_batch`` is the CIFAR-10-shaped synthetic batch
of SURVEY.md 8(d) (generated on CPU so the CPU oracle and the GPU see identical bits).
"""
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

from micronet_amd.models import nin, nin_gc, resnet


def init_like_main(model):
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, 0, 0.01)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
    return model


def build_model(arch, seed=1):
    torch.manual_seed(seed)
    ctor = {"nin": nin.Net, "nin_gc": nin_gc.Net, "resnet18": resnet.resnet18, "resnet34": resnet.resnet34,
            "resnet50": resnet.resnet50}[arch]
    return init_like_main(ctor())


def synth_batch(batch, seed=1234, device=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(batch, 3, 32, 32, generator=g)
    y = torch.randint(0, 10, (batch,), generator=g)
    if device is not None:
        x, y = x.to(device), y.to(device)
    return x, y


def make_optimizer(model, lr=0.01, weight_decay=1e-5, fused=None, **kw):
    """Adam with one parameter group per tensor, as main.py:308-315 builds it.  On the GPU the step runs as ONE launch
    over all tensors (micronet_amd.optim.Adam, same update and state layout as torch.optim.Adam); ``fused=False`` keeps
    torch's implementation (used by tests to compare)."""
    groups = [{"params": [p], "lr": lr, "weight_decay": weight_decay} for _, p in model.named_parameters()]
    if fused is None:
        fused = all(p.is_cuda for g in groups for p in g["params"])
    if fused:
        from micronet_amd.optim import Adam
        return Adam(groups, lr=lr, weight_decay=weight_decay, **kw)
    return torch.optim.Adam(groups, lr=lr, weight_decay=weight_decay, **kw)


def train_step(model, optimizer, data, target):
    output = model(data)
    loss = cross_entropy(output, target)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss, output


def cross_entropy(output, target):
    """``nn.CrossEntropyLoss()`` of the reference's main.py (wqaq/dorefa/main.py:87-92): the loss and its gradient in one launch on the GPU (ops.CrossEntropy), ATen's
    otherwise"""
    from micronet_amd import ops
    return ops.cross_entropy(output, target)


def bump_bn_counters(model):
    """``num_batches_tracked += 1`` of every BatchNorm the streaming kernels run (``BatchNorm2dReLU`` / ``BatchNorm2dPlain`` on their plain paths: 20 tiny launches per
    resnet18 step) as ONE multi-tensor launch in front of the forward; each module is told that its counter already moved for the coming forward (the flag is
    consumed there).  Modules with ``momentum=None`` read the counter on the host and keep doing their own increment."""
    from micronet_amd.quantization.wqaq.dorefa.quantize import BatchNorm2dPlain, BatchNorm2dReLU
    from micronet_amd.nn import TailBNMixin
    todo = []
    for m in model.modules():
        if isinstance(m, (BatchNorm2dReLU, BatchNorm2dPlain, TailBNMixin)) and m.training and m.track_running_stats and m.num_batches_tracked is not None \
                and m.momentum is not None and m.num_batches_tracked.is_cuda and not getattr(m, "q_out_bits", 0) and "_mn_nbt_pre" not in m.__dict__:
            todo.append(m)
    if len(todo) >= 2:
        torch._foreach_add_([m.num_batches_tracked for m in todo], 1)
        for m in todo:
            m.__dict__["_mn_nbt_pre"] = True


def prefetch_weight_path(model, late=None):
    """``late`` (two-bucket data-parallel step, GraphedTrainStep): ids of the modules behind the bucket boundary -- every multi-tensor weight node is then built
    per bucket, so that the late bucket's weight gradients exist when the first half of backward ends (one node over all weights would run last).

    Quantize the weights of every W-ternary conv of ``model`` NOW, in one launch (ops.MultiTernaryWeight: one autograd node, so the
    backward is one launch too); the owning conv picks its tensor up in its forward.  A step of nin_gc saves 12 launches of ~5 us.
    (A second stream for this path was measured slower under graph replay -- the fork / join edges cost more than the launches they hide -- and is gone.)

    CONTRACT for IAO nets: the prefetch performs the weight OBSERVER's step of this iteration (EMA of the per-channel range, ``num_flag``, scale / zero-point) for
    every module it covers -- the reference performs it inside the module's forward (iao/quantize.py:214-221).  Each call must therefore be followed by exactly
    one forward of every covered module before the next call; a second prefetch that finds an unconsumed hand-over raises instead of advancing the observers twice."""
    from micronet_amd import ops
    from micronet_amd.quantization.wbwtab import quantize as wb
    from micronet_amd.quantization.wqaq.dorefa import quantize as dr
    # DoReFa nets: every conv / linear weight quantizer of one bit-width in one MultiDorefaWeight node (2 launches forward, 3 backward per step)
    by_bits = {}
    for m in model.modules():
        if isinstance(m, (dr.QuantConv2d, dr.QuantLinear)) and not m.quant_inference and 2 <= m.weight_quantizer.w_bits <= 31 and m.weight.is_cuda \
                and m.weight.is_contiguous():
            m.weight_quantizer.__dict__.pop("_mn_pre", None)
            by_bits.setdefault((m.weight_quantizer.w_bits, bool(late) and id(m) in late), []).append(m)
    for (bits, _), ms in by_bits.items():
        for i in range(0, len(ms), 32):
            grp = ms[i:i + 32]
            if len(grp) < 2:
                continue
            qws = ops.MultiDorefaWeight.apply(bits, *[m.weight for m in grp])
            for m, wq in zip(grp, qws):
                m.weight_quantizer._mn_pre = (m.weight, wq, None)
            if 2 <= bits <= 8:          # the dense layers (ResNets): their weight codes in both fragment orders, one launch for the net
                ops.pack_dense_weights([(m, wq) for m, wq in zip(grp, qws) if isinstance(m, dr.QuantConv2d)], bits)
                # ... and the pointwise layers of the fused k-bit blocks (nin_gc): forward / backward-data images, one launch
                ops.pack_pointwise_weights([(m, wq) for m, wq in zip(grp, qws) if isinstance(m, dr.QuantConv2d)], (ops.WQ_DOREFA, bits, 0, 0, None))
    # IAO nets: every per-channel weight quantizer (observer update + qparams + fake-quant of each output channel) of one flavour in one MultiIaoWeight
    # node, then the dense layers' weight codes in one launch.  QuantBNFuseConv2d is excluded: it quantises weights folded with THIS step's batch statistics.
    from micronet_amd.quantization.wqaq.iao import quantize as ia
    groups = {}
    for m in model.modules():
        if type(m) in (ia.QuantConv2d, ia.QuantLinear) and not m.quant_inference and m.training and m.weight.is_cuda and m.weight.is_contiguous() \
                and m.weight.dtype == torch.float32:
            q = m.weight_quantizer
            obs = q.observer
            stale = q.__dict__.pop("_mn_pre", None)
            if stale is not None and stale[0] is m.weight:
                raise RuntimeError("prefetch_weight_path: the IAO weight hand-over of the previous call was never consumed (a covered module did not run its "
                                   "forward since); its observer would advance twice in one iteration -- see the contract in this function's docstring")
            if 2 <= q.bits <= 24 and not q.qaft and getattr(obs, "q_level", None) in ("C", "FC") and getattr(obs, "_kind", None) in (0, 1) \
                    and not getattr(obs, "_mn_sync", False) and obs.min_val.numel() == m.weight.shape[0]:
                groups.setdefault((q.bits, q._q_type_static, obs._kind, float(getattr(obs, "momentum", 0.1)), bool(late) and id(m) in late), []).append(m)
    for cfg, ms in groups.items():
        cfg = cfg[:4]
        for i in range(0, len(ms), 32):
            grp = ms[i:i + 32]
            if len(grp) < 2:
                continue
            state = []
            for m in grp:
                q, obs = m.weight_quantizer, m.weight_quantizer.observer
                qp = torch.empty((m.weight.shape[0], 4), dtype=torch.float32, device=m.weight.device)
                state.append((obs.min_val, obs.max_val, q.scale, q.zero_point, qp, obs.num_flag == 0))
            qws = ops.MultiIaoWeight.apply(cfg, state, *[m.weight for m in grp])
            for m, wq, st in zip(grp, qws, state):
                obs = m.weight_quantizer.observer
                if obs.num_flag == 0:
                    obs.num_flag += 1
                m.weight_quantizer._mn_pre = (m.weight, wq, st[4])
            if cfg[1] == 0 and 2 <= cfg[0] <= 8:
                convs = [(m, wq, st[4]) for m, wq, st in zip(grp, qws, state) if type(m) is ia.QuantConv2d]
                ops.pack_dense_weights([(m, wq) for m, wq, _ in convs], cfg[0], qps=[qp for _, _, qp in convs])
    mods = [m for m in model.modules() if isinstance(m, wb.QuantConv2d) and not m.quant_inference and m.weight_quantizer.W in (2, 3)]
    for m in mods:
        m.weight_quantizer.__dict__.pop("_mn_pre", None)
    tern = [m for m in mods if m.weight_quantizer.W == 3 and m.weight.is_cuda and m.weight.is_contiguous()]
    if late:
        tern = [m for m in tern if id(m) not in late] + [m for m in tern if id(m) in late]
    bounds = [0, sum(1 for m in tern if id(m) not in late), len(tern)] if late else [0, len(tern)]
    for lo_, hi_ in zip(bounds[:-1], bounds[1:]):
      for i in range(lo_, hi_, 32):
        grp = tern[i:min(i + 32, hi_)]
        if len(grp) < 2:
            continue
        qws = ops.MultiTernaryWeight.apply(*[m.weight for m in grp])
        for m, wq in zip(grp, qws):
            m.weight_quantizer._mn_pre = (m.weight, wq, None)
        ops.pack_pointwise_weights(list(zip(grp, qws)), (ops.WQ_TERNARY, 0, 0, 0, None))          # the pointwise blocks' code images: one launch for the net
    # W = 2 (binary weights, the literal `--W 2 --A 2` of wbwtab/main.py): the same two launches (round 6: until then 7 + 7 quantizer launches and, the codes not
    # existing yet at the start of the step, 10 per-call packs -- 186 us of a 1.9 ms step).  The in-place mean-centring of every weight happens here, before the
    # first forward of the step instead of inside each conv's forward: the same values either way (no layer reads another layer's weight).
    bins = [m for m in mods if m.weight_quantizer.W == 2 and m.weight.is_cuda and m.weight.is_contiguous() and m.weight.dim() == 4 and m.weight.shape[2] * m.weight.shape[3] <= 256]
    if late:
        bins = [m for m in bins if id(m) not in late] + [m for m in bins if id(m) in late]
    bounds = [0, sum(1 for m in bins if id(m) not in late), len(bins)] if late else [0, len(bins)]
    for lo_, hi_ in zip(bounds[:-1], bounds[1:]):
      for i in range(lo_, hi_, 32):
        grp = bins[i:min(i + 32, hi_)]
        if len(grp) < 2:
            continue
        qws = ops.MultiBinaryWeight.apply(*[m.weight for m in grp])
        for m, wq in zip(grp, qws):
            m.weight_quantizer._mn_pre = (m.weight, wq, None)
        ops.pack_pointwise_weights(list(zip(grp, qws)), (ops.WQ_TERNARY, 0, 0, 0, None))
    return


def pick_bucket_boundary(model):
    """Where the data-parallel gradient exchange of the graphed step is split in two: a top-level stage of ``model`` such that the parameters BEHIND it (the late
    bucket: backward produces their gradients first) can be all-reduced while the stages up to it are still in backward.  Returns (boundary module, [late
    parameters], [early parameters]) or None (one bucket).  MN_DP_BUCKETS=1 forces one bucket, =2 asks for two whatever the size; the default is two from 16 MB of
    gradients (resnet18: 44.7 MB -- the boundary falls behind conv3_x: 42 MB overlap the remaining half of backward, 2.7 MB stay exposed; nin_gc's 2.4 MB: one)."""
    import os
    env = os.environ.get("MN_DP_BUCKETS", "")
    params = [p for p in model.parameters() if p.requires_grad]
    total = sum(p.numel() * p.element_size() for p in params)
    if env == "1" or (env != "2" and total < (16 << 20)):
        return None
    root = model
    while True:                                   # descend through wrappers that hold everything in one child (nin_gc: Net.tnn_bin)
        kids = [k for k in root.children()]
        with_p = [k for k in kids if any(p.requires_grad for p in k.parameters())]
        if len(with_p) == 1 and len(list(with_p[0].children())) > 1:
            root = with_p[0]
        else:
            break
    if len(kids) < 3:
        return None
    convs = [sum(1 for m in k.modules() if isinstance(m, (nn.Conv2d, nn.Linear))) for k in kids]
    nconv = max(1, sum(convs))
    for i in range(len(kids) - 1):                # the earliest boundary that leaves >= 40 % of the weight layers (the work of the second half of backward) in front of it
        if sum(convs[:i + 1]) >= 0.4 * nconv and sum(convs[i + 1:]) > 0:
            early_ids = {id(p) for k in kids[:i + 1] for p in k.parameters()}
            early = [p for p in params if id(p) in early_ids]
            late = [p for p in params if id(p) not in early_ids]
            if early and late:
                return kids[i], late, early
            return None
    return None


# Capture mode of every HIP-graph capture in this module: "thread_local" -- only THIS thread's unsafe calls invalidate a capture.  With a process group alive,
# ProcessGroupNCCL's watchdog thread polls the events of finished collectives (hipEventQuery) on its own schedule; under the default "global" mode one such poll
# while a capture is open is "operation not permitted when stream is capturing" and takes the process down (seen in bench.py's dp_single_rank leg, round 6).
_CAPTURE_MODE = "thread_local"


class GraphedTrainStep:
    """The training step of ``train_step`` captured ONCE in HIP graphs and replayed: ~150 kernel launches per nin_gc step
    become one ``hipGraphLaunch``, so the host never gates the GPU (the eager step spends as long in Python / ctypes /
    autograd bookkeeping as the kernels take).

    world == 1: one graph = forward + loss + zero_grad + backward + Adam.
    world  > 1: graph A = forward + loss + zero_grad + backward + packing of all gradients into one flat bucket (already
    divided by the world size); the RCCL all-reduce of that bucket runs eagerly on the same stream; graph B = Adam reading
    the reduced bucket.  (nin_gc: one 2.4 MB collective per step.)  From 16 MB of gradients (resnet18) the step has TWO buckets:
    graph A1 ends where backward reaches the output of a boundary stage (``pick_bucket_boundary``), the late bucket's all-reduce
    overlaps graph A2 (the rest of backward), the small early bucket follows, then graph B.  IAO models whose activation observers reduce their
    range over the ranks (dp.sync_observers) are captured in SEGMENTS cut at those collectives -- ``self.segments`` --
    and replayed as segment, range collective, segment, ..., graph A, gradient all-reduce, graph B.

    Data is fed through the static tensors ``self.data`` / ``self.target`` (``copy_`` new batches into them); ``self.loss`` /
    ``self.output`` hold the results of the last replay.  The optimizer must be ``micronet_amd.optim.Adam``: its step count and every
    group's ``lr`` / ``weight_decay`` live in device memory, and ``step()`` refreshes them from ``param_groups`` before each replay, so
    the reference's per-epoch ``adjust_learning_rate`` (wbwtab/main.py:62-66, 343) keeps working on a replayed step.

    What capture freezes (evaluated once, at construction): the module graph and every Python-side branch in it -- train/eval mode,
    IAO observers' first-call flag (the capture happens after ``warmup`` real steps, so observers are past their first call),
    ``BatchNorm.momentum is None``, Adam's betas / eps (changing them raises).  Construction itself CONSUMES ``warmup`` (default 3)
    real training steps on ``data`` / ``target`` -- parameters, Adam moments, BN running statistics and IAO observer state advance
    exactly as in three eager steps; count them in the schedule.  ``optimizer.state_dict()`` syncs the step count automatically."""

    def __init__(self, model, optimizer, data, target, warmup=3, group=None):
        import torch.distributed as dist
        self.model, self.optimizer = model, optimizer
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        if not data.is_cuda:
            raise RuntimeError("GraphedTrainStep: HIP graphs need device tensors (got %s); use the eager DP step (dp.train_step_dp)" % data.device)
        from micronet_amd import dp
        self.dp = dp.active(group)            # gradient all-reduce between graph A and graph B (more than one rank, or the MN_DP_SINGLE=1 measurement mode)
        self._host_sync = self.dp and dist.get_backend(group) != "nccl"
        self.data, self.target = data.clone(), target.clone()
        self.params = [p for p in model.parameters() if p.requires_grad]
        # Synced IAO observers (dp.sync_observers) put one 2-float MAX all-reduce per activation quantizer INSIDE forward, each needed before the next layer can
        # run.  The step is then captured in SEGMENTS: the capture is cut at every such collective (dp.allreduce_minmax -> self._cut), a replayed step is
        # segment, collective, segment, ... with the last segment holding the rest of forward, the loss and the whole backward.  The collectives stay ordinary
        # torch.distributed calls between replays (any backend: RCCL enqueues them on the device behind the segment, no host round trip), so nothing depends on
        # the backend being able to record a collective into a graph.
        self.segments = []                    # [(graph, operand of the collective that follows it, group)]
        self._segmented = self.dp and any(getattr(m, "_mn_sync", False) for m in model.modules())
        # Two gradient buckets (pick_bucket_boundary): graph A1 = forward + loss + backward down to the boundary stage's output + packing of the late bucket; its
        # all-reduce is issued asynchronously and overlaps graph A2 = the rest of backward + packing of the early bucket; that bucket's all-reduce; graph B = Adam.
        # The boundary is made an autograd LEAF (forward hook: the stage's output is replaced by its detached twin), so each half is an ordinary backward call.
        self.bound = pick_bucket_boundary(model) if self.dp else None
        self._late_mods = None
        if self.bound is not None:
            late_ids = {id(p) for p in self.bound[1]}
            self._late_mods = {id(m) for m in model.modules() if any(id(p) in late_ids for p in m.parameters(recurse=False))}
        self.graph_a2, self.flat2, self.one_bucket_reason = None, None, None
        if not hasattr(optimizer, "capturable"):
            raise TypeError("GraphedTrainStep needs micronet_amd.optim.Adam")
        optimizer.capturable = True
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):           # eager warm-up on a side stream: allocator, MIOpen solver search, LDS attributes
                self._fwd_bwd()
                self._reduce_eager()
                optimizer.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph_b = None
        self.flat = None
        import gc
        from micronet_amd import dp
        gc.collect()
        torch.cuda.empty_cache()
        pool = torch.cuda.graph_pool_handle()
        cap = torch.cuda.Stream()             # ONE capture stream for all segments: backward's nodes run on the stream their forward ran on
        cap.wait_stream(torch.cuda.current_stream())
        self.graph_a = torch.cuda.CUDAGraph()
        self._capturing = self.graph_a
        try:
            with torch.cuda.stream(cap):
                self._capturing.capture_begin(pool=pool, capture_error_mode=_CAPTURE_MODE)
                try:
                    if self._segmented:
                        dp._segment_cut = lambda buf, group: self._cut(buf, group, pool)
                    self._fwd_bwd(mid=(lambda: self._cut_backward(pool)) if self.bound is not None else None)
                    if not self.dp:
                        optimizer.step()
                    elif self.bound is not None:
                        self.flat2 = torch.cat([p.grad.reshape(-1) for p in self.bound[2]])
                        self.flat2.div_(self.world)
                    else:
                        self.flat = torch.cat([p.grad.reshape(-1) for p in self.params])
                        self.flat.div_(self.world)
                except BaseException:
                    self.segments.clear()                 # (graphs captured so far are dropped with their pool references)
                    raise
                finally:
                    dp._segment_cut = None
                    try:
                        self._capturing.capture_end()
                    except Exception:                      # noqa: BLE001 -- a cut that failed half-way leaves no open capture: the ORIGINAL error must surface
                        if sys.exc_info()[0] is None:
                            raise
            if self.bound is not None:
                self.graph_a2 = self._capturing    # (graph_a was closed by _cut_backward)
            else:
                self.graph_a = self._capturing    # the LAST segment (the only one without range collectives)
        except Exception as e:          # noqa: BLE001 -- a failed capture must leave a usable process behind
            torch.cuda.synchronize()
            optimizer.capturable = False
            raise RuntimeError("GraphedTrainStep: capture failed (%s: %s); use the eager step" % (type(e).__name__, str(e)[:200])) from e
        torch.cuda.current_stream().wait_stream(cap)
        if self.dp:
            self._captured_grads = [p.grad for p in self.params]   # graph A writes these on every replay: keep them allocated
            for flat, ps in ((self.flat, self.params),) if self.bound is None else ((self.flat, self.bound[1]), (self.flat2, self.bound[2])):
                off = 0
                for p in ps:                  # Adam reads the reduced buckets in place
                    p.grad = flat[off:off + p.numel()].view_as(p)
                    off += p.numel()
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, pool=self.graph_a.pool(), capture_error_mode=_CAPTURE_MODE):
                optimizer.step()

    def _fwd_bwd(self, mid=None):
        prefetch_weight_path(self.model, late=self._late_mods)   # all weight quantizers of the step in one launch (and one in backward) per gradient bucket
        bump_bn_counters(self.model)
        if self.bound is None:
            self.output = self.model(self.data)
            self.loss = cross_entropy(self.output, self.target)
            self.optimizer.zero_grad(set_to_none=True)
            self.loss.backward()
            return
        cutpt = {}

        def leaf(mod, inp, out):          # the boundary stage's output becomes a leaf: backward stops there, the second call continues from it
            if type(out) is not torch.Tensor or not out.requires_grad:
                cutpt["bad"] = type(out).__name__                # a packed hand-over (QActTensor / SignTensor: codes + lazily expanded gradients cross this edge): one bucket
                return None
            cutpt["out"] = out
            cutpt["leaf"] = out.detach().requires_grad_(True)
            return cutpt["leaf"]
        h = self.bound[0].register_forward_hook(leaf)
        try:
            self.output = self.model(self.data)
        finally:
            h.remove()
        self.loss = cross_entropy(self.output, self.target)
        self.optimizer.zero_grad(set_to_none=True)
        if "leaf" not in cutpt:
            self.bound, self._late_mods, self.one_bucket_reason = None, None, "boundary output is a %s" % cutpt.get("bad", "tensor outside the graph")
            self.loss.backward()
            return
        self.loss.backward()                                  # gradients of the late bucket + of the boundary leaf
        if mid is not None:
            mid()
        g = cutpt["leaf"].grad
        cutpt["leaf"].grad = None
        cutpt["out"].backward(g)                              # the early bucket

    def _cut_backward(self, pool):
        """Capture only: pack the late bucket behind the first half of backward, end graph A1 and begin graph A2."""
        self.flat = torch.cat([p.grad.reshape(-1) for p in self.bound[1]])
        self.flat.div_(self.world)
        self._capturing.capture_end()
        self.graph_a = self._capturing
        self._capturing = torch.cuda.CUDAGraph()
        self._capturing.capture_begin(pool=pool, capture_error_mode=_CAPTURE_MODE)

    def _reduce_eager(self):
        if self.dp:
            import torch.distributed as dist
            for ps in ((self.params,) if self.bound is None else (self.bound[1], self.bound[2])):
                flat = torch.cat([p.grad.reshape(-1) for p in ps]).div_(self.world)
                dist.all_reduce(flat, group=self.group)
                off = 0
                for p in ps:
                    p.grad = flat[off:off + p.numel()].view_as(p)
                    off += p.numel()

    def _cut(self, buf, group, pool):
        """End the segment being captured at a range collective on ``buf`` (allocated by the segment: a fixed address of the graphs' pool) and begin the next."""
        self._capturing.capture_end()
        self.segments.append((self._capturing, buf, group))
        self._capturing = torch.cuda.CUDAGraph()
        self._capturing.capture_begin(pool=pool, capture_error_mode=_CAPTURE_MODE)

    def step(self):
        self.optimizer.refresh_hyper()        # lr / weight_decay edits of the training loop reach the captured Adam launch
        if self.dp:
            import torch.distributed as dist
            # gloo only (the functional check of this path on one GPU): its collective on a device tensor does not order itself behind a freshly launched graph,
            # hence the host synchronisation.  RCCL enqueues each collective behind the preceding replay on the device (event wait on the current stream) and
            # the next replay behind the collective: no host round trip in the step.
            for g, buf, group in self.segments:
                g.replay()
                if self._host_sync:
                    torch.cuda.current_stream().synchronize()
                dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=group)
            self.graph_a.replay()
            if self._host_sync:
                torch.cuda.current_stream().synchronize()
            if self.graph_a2 is None:
                dist.all_reduce(self.flat, group=self.group)
            else:
                # RCCL: the late bucket's all-reduce runs on the communicator's stream behind graph A1 (async_op) while graph A2 -- the rest of backward -- replays
                # on this one; both collectives are joined in front of graph B
                w1 = dist.all_reduce(self.flat, group=self.group, async_op=True)
                self.graph_a2.replay()
                if self._host_sync:
                    torch.cuda.current_stream().synchronize()
                w2 = dist.all_reduce(self.flat2, group=self.group, async_op=True)
                w1.wait()
                w2.wait()
            rb = getattr(self.model, "_mn_replica_buffers", None)
            if rb is not None:          # the reference's DataParallel buffer semantics (dp.ReplicaBuffers): rank 0's observer / BatchNorm state, one flat broadcast per step
                rb.exchange()
            self.graph_b.replay()
        else:
            self.graph_a.replay()
        return self.loss, self.output

    def finish(self):
        """Bring host-side optimizer state (step counts) up to date with the replays."""
        self.optimizer.sync_steps()
