"""On-device CIFAR-10 input pipeline -- the caller side of the hot path.

The reference feeds the step from a 2-worker CPU ``DataLoader`` over torchvision transforms (wqaq/dorefa/main.py:203-236: RandomCrop(32, padding=4),
RandomHorizontalFlip, ToTensor, Normalize((0.4914, 0.4822, 0.4465), (0.2023, 0.1994, 0.2010))).  At the 80-100 k images/s of the fused QAT step that
loader is two orders of magnitude too slow, so here the uint8 dataset (150 MB) lives in HBM and every batch is gathered + augmented by ONE gfx950
kernel (``mn_cifar_augment``); the random draws (epoch permutation, crop offsets, flips) come from a ``torch.Generator`` on the device, so an epoch is
reproducible from its seed and the kernel is bit-identical to the CPU transforms for the same draws (tests/test_gpu_data.py).

``save_state`` / ``load_state`` write and read the checkpoint layout of the reference scripts (main.py:32-59: ``{"best_acc", "state_dict"}`` with any
``module.`` prefix stripped), so checkpoints interchange with the reference's ``--resume`` / ``--refine`` paths."""
import ctypes as C

import torch

from . import _lib

CIFAR_MEAN = (0.4914, 0.4822, 0.4465)
CIFAR_STD = (0.2023, 0.1994, 0.2010)


def augment(images_u8, index, ox, oy, flip, pad=4, mean=CIFAR_MEAN, std=CIFAR_STD):
    """images_u8: uint8 [n][H][W][C] on the GPU; index / ox / oy: int32 [B]; flip: uint8 [B] -> float32 [B][C][H][W]."""
    if not images_u8.is_cuda or images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or not images_u8.is_contiguous():
        raise _lib.MicronetHipError("augment: images must be a contiguous uint8 [n][H][W][C] CUDA tensor (no CPU fallback)")
    n, H, W, Cc = images_u8.shape
    B = index.numel()
    for t, dt in ((index, torch.int32), (ox, torch.int32), (oy, torch.int32), (flip, torch.uint8)):
        if t.dtype != dt or not t.is_cuda or t.numel() != B or not t.is_contiguous():
            raise _lib.MicronetHipError("augment: index / ox / oy must be int32 [B], flip uint8 [B], all on the GPU")
    out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=images_u8.device)
    lib = _lib.get_lib()
    FA = C.c_float * Cc
    with torch.cuda.device(images_u8.device):
        rc = lib.mn_cifar_augment(C.c_void_p(images_u8.data_ptr()), n, C.c_void_p(index.data_ptr()), C.c_void_p(ox.data_ptr()), C.c_void_p(oy.data_ptr()),
                                  C.c_void_p(flip.data_ptr()), B, H, W, Cc, int(pad), FA(*mean), FA(*std), C.c_void_p(out.data_ptr()),
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        lib.check(rc, "mn_cifar_augment")
    return out


class DeviceCifarLoader:
    """Iterates (data, target) batches of one epoch like ``DataLoader(trainset, batch_size, shuffle=True)`` over the training transforms, entirely on
    the GPU.  ``images``: uint8 [n][32][32][3] (torchvision ``CIFAR10.data``), ``labels``: int64 [n].  ``train=False``: no crop / flip / shuffle (the
    test transform: ToTensor + Normalize)."""

    def __init__(self, images, labels, batch_size, train=True, seed=1, device="cuda", drop_last=False, pad=4, mean=CIFAR_MEAN, std=CIFAR_STD):
        self.images = torch.as_tensor(images, dtype=torch.uint8).contiguous().to(device)
        self.labels = torch.as_tensor(labels, dtype=torch.int64).to(device)
        self.batch_size, self.train, self.drop_last, self.pad, self.mean, self.std = int(batch_size), train, drop_last, pad, mean, std
        self.gen = torch.Generator(device=self.images.device).manual_seed(seed)

    def __len__(self):
        n = self.images.shape[0]
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def draw(self, B):
        """the random draws of one batch: crop offsets uniform in [0, 2 pad] (RandomCrop.get_params), flips with p = 0.5"""
        dev = self.images.device
        if not self.train:
            z = torch.full((B,), self.pad, dtype=torch.int32, device=dev)
            return z, z.clone(), torch.zeros(B, dtype=torch.uint8, device=dev)
        ox = torch.randint(0, 2 * self.pad + 1, (B,), generator=self.gen, device=dev, dtype=torch.int32)
        oy = torch.randint(0, 2 * self.pad + 1, (B,), generator=self.gen, device=dev, dtype=torch.int32)
        flip = (torch.rand(B, generator=self.gen, device=dev) < 0.5).to(torch.uint8)
        return ox, oy, flip

    def __iter__(self):
        n = self.images.shape[0]
        dev = self.images.device
        order = torch.randperm(n, generator=self.gen, device=dev) if self.train else torch.arange(n, device=dev)
        for i in range(len(self)):
            idx = order[i * self.batch_size:(i + 1) * self.batch_size]
            ox, oy, flip = self.draw(idx.numel())
            yield augment(self.images, idx.to(torch.int32).contiguous(), ox, oy, flip, self.pad, self.mean, self.std), self.labels[idx]


def save_state(model, best_acc, path, cfg=None):
    """The checkpoint layout of the reference scripts (wqaq/dorefa/main.py:32-59)."""
    sd = {k.replace("module.", ""): v for k, v in model.state_dict().items()}
    state = {"best_acc": best_acc, "state_dict": sd}
    if cfg is not None:
        state["cfg"] = cfg
    torch.save(state, path)


def load_state(model, path, strict=True):
    state = torch.load(path, map_location="cpu")
    sd = {k.replace("module.", ""): v for k, v in state["state_dict"].items()}
    model.load_state_dict(sd, strict=strict)
    return state.get("best_acc", 0.0)
