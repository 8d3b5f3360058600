"""``Adam``: drop-in for ``torch.optim.Adam`` as the reference training scripts construct it (``*/main.py:308-315``: one
parameter group per tensor, ``lr``, ``weight_decay``), executed as ONE gfx950 launch over all tensors (``mn_adam_step``)
instead of ~7 small kernels per group.  Same update as torch (amsgrad off, L2 weight decay folded into the gradient),
same ``state_dict`` layout (``step``, ``exp_avg``, ``exp_avg_sq``), so checkpoints interchange.

``capturable = True`` keeps the step count AND every group's ``lr`` / ``weight_decay`` in device memory (``mn_adam_step_dev``) so that
``step()`` can be captured in a HIP graph and replayed (micronet_amd.train.GraphedTrainStep): ``refresh_hyper()`` -- called before every
replay -- copies the groups' current ``lr`` / ``weight_decay`` into that device table when the training loop edited them (the reference's
``adjust_learning_rate``, wbwtab/main.py:62-66), ``sync_steps()`` writes the device step count back into ``state``.  betas / eps are
frozen at capture time; changing them afterwards raises."""
import ctypes as C

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.capturable = False
        self._step_dev = None
        self._hyper_dev = {}          # (betas, eps) batch key -> device [n][2] = lr, weight_decay
        self._hyper_host = {}

    def _host_step(self):
        steps = {int(st["step"]) for st in self.state.values() if st}
        if len(steps) > 1:
            raise _lib.MicronetHipError("capturable Adam needs one common step count for all parameters")
        return steps.pop() if steps else 0

    def sync_steps(self):
        """Copy the device-side step count (advanced by graph replays) into every ``state[p]['step']``."""
        if self._step_dev is not None:
            n = int(self._step_dev.item())
            for st in self.state.values():
                if st:
                    st["step"].fill_(n)

    def state_dict(self):
        self.sync_steps()             # a checkpoint taken between graph replays carries the replayed step count
        return super().state_dict()

    def _hyper_items(self):
        items = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                items.setdefault((float(b1), float(b2), float(group["eps"])), []).append((float(group["lr"]), float(group["weight_decay"])))
        return items

    def refresh_hyper(self):
        """Bring the device-side {lr, weight_decay} table up to date with ``param_groups`` (no-op when nothing changed; one small
        host-to-device copy per batch key when the schedule moved).  Raises if betas / eps differ from the captured values."""
        if not self._hyper_dev:
            return
        items = self._hyper_items()
        if set(items) != set(self._hyper_dev):
            raise _lib.MicronetHipError("capturable Adam: betas / eps changed after the step was captured; re-capture the step")
        for key, vals in items.items():
            if vals != self._hyper_host[key]:
                if len(vals) != len(self._hyper_host[key]):
                    raise _lib.MicronetHipError("capturable Adam: parameter groups changed after capture")
                self._hyper_dev[key].copy_(torch.tensor(vals, dtype=torch.float32), non_blocking=False)
                self._hyper_host[key] = vals

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.get_lib()
        if self.capturable:
            return self._step_capturable(lib, loss)
        # tensors that share (step, betas, eps) go into one launch table
        batches = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_cuda:
                    raise _lib.MicronetHipError("micronet_amd.optim.Adam handles dense float32 CUDA parameters")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if not p.is_contiguous():
                    raise _lib.MicronetHipError("non-contiguous parameter")
                key = (int(st["step"]), float(b1), float(b2), float(group["eps"]), p.device.index)
                batches.setdefault(key, []).append((p, g, st, float(group["lr"]), float(group["weight_decay"])))
        for (step, b1, b2, eps, dev), items in batches.items():
            arr = (_lib.AdamTensor * len(items))()
            for i, (p, g, st, lr, wd) in enumerate(items):
                arr[i] = _lib.AdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), lr, wd)
            with torch.cuda.device(dev):
                rc = lib.mn_adam_step(arr, len(items), step, b1, b2, eps, C.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                lib.check(rc, "mn_adam_step")
        return loss

    def _step_capturable(self, lib, loss):
        items, dev = {}, None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    raise _lib.MicronetHipError("capturable Adam: every parameter needs a gradient on every step")
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise _lib.MicronetHipError("micronet_amd.optim.Adam handles dense contiguous float32 CUDA parameters")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                dev = p.device
                items.setdefault((float(b1), float(b2), float(group["eps"])), []).append(
                    (p, p.grad, st, float(group["lr"]), float(group["weight_decay"])))
        if dev is None:
            return loss
        if self._step_dev is None:
            self._step_dev = torch.full((1,), self._host_step(), dtype=torch.int32, device=dev)
        self._step_dev.add_(1)
        for key, its in items.items():
            b1, b2, eps = key
            vals = [(lr, wd) for (_, _, _, lr, wd) in its]
            if key not in self._hyper_dev:          # first capturable step (eager warm-up, outside any capture): allocate the table
                self._hyper_dev[key] = torch.tensor(vals, dtype=torch.float32, device=dev)
                self._hyper_host[key] = vals
            elif vals != self._hyper_host[key] and not torch.cuda.is_current_stream_capturing():
                self._hyper_dev[key].copy_(torch.tensor(vals, dtype=torch.float32))
                self._hyper_host[key] = vals
            arr = (_lib.AdamTensor * len(its))()
            for i, (p, g, st, lr, wd) in enumerate(its):
                arr[i] = _lib.AdamTensor(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), lr, wd)
            with torch.cuda.device(dev):
                rc = lib.mn_adam_step_dev(arr, len(its), C.c_void_p(self._step_dev.data_ptr()), C.c_void_p(self._hyper_dev[key].data_ptr()), b1, b2, eps,
                                          C.c_void_p(torch.cuda.current_stream().cuda_stream))
            if rc != 0:
                lib.check(rc, "mn_adam_step_dev")
        return loss
