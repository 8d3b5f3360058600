"""Drive the C ABI with raw pointers from either numpy arrays (CPU emulation build) or torch CUDA tensors (gfx950 build).

Test helper only.  ``Backend('emu')`` loads tests/emu/build/libmicronet_emu.so and keeps data in numpy;
``Backend('gpu')`` loads the product library and keeps data in torch CUDA tensors.  Both go through the same
``micronet_amd._lib`` prototypes, so the parity tests below are written once.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from micronet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_SO = os.path.join(ROOT, "tests", "emu", "build", "libmicronet_emu.so")


def build_emu():
    srcs = [os.path.join(ROOT, "micronet_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "micronet_amd", "csrc"))]
    srcs.append(os.path.join(ROOT, "tests", "emu", "include", "hip", "hip_runtime.h"))
    if not os.path.exists(EMU_SO) or any(os.path.getmtime(s) > os.path.getmtime(EMU_SO) for s in srcs):
        subprocess.check_call([os.path.join(ROOT, "tests", "emu", "build_emu.sh")])
    return EMU_SO


class Backend:
    def __init__(self, kind):
        self.kind = kind
        if kind == "emu":
            self.lib = _lib.load(build_emu())
            assert self.lib.mn_is_emulation() == 1
            self.stream = None
        else:
            import torch
            self.torch = torch
            self.lib = _lib.get_lib()
            self.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- buffers
    def to_dev(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        if self.kind == "emu":
            return a.copy()
        return self.torch.from_numpy(a.copy()).cuda()

    def to_dev_i8(self, a):
        a = np.ascontiguousarray(a, dtype=np.int8)
        if self.kind == "emu":
            return a.copy()
        return self.torch.from_numpy(a.copy()).cuda()

    def to_dev_u8(self, a):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        if self.kind == "emu":
            return a.copy()
        return self.torch.from_numpy(a.copy()).cuda()

    def to_dev_i16(self, a):
        a = np.ascontiguousarray(a, dtype=np.int16)
        if self.kind == "emu":
            return a.copy()
        return self.torch.from_numpy(a.copy()).cuda()

    def to_dev_i64(self, a):
        a = np.ascontiguousarray(a, dtype=np.int64)
        if self.kind == "emu":
            return a.copy()
        return self.torch.from_numpy(a.copy()).cuda()

    def empty_i8(self, shape):
        if self.kind == "emu":
            return np.full(shape, 77, dtype=np.int8)   # poison
        return self.torch.full(tuple(shape), 77, dtype=self.torch.int8, device="cuda")

    def empty(self, shape):
        if self.kind == "emu":
            return np.full(shape, np.float32(-1234.5), dtype=np.float32)   # poison
        return self.torch.full(tuple(shape) if not isinstance(shape, int) else (shape,), -1234.5, dtype=self.torch.float32, device="cuda")

    def to_host(self, b):
        if self.kind == "emu":
            return np.array(b, copy=True)
        self.torch.cuda.synchronize()
        return b.detach().cpu().numpy()

    def ptr(self, b):
        if b is None:
            return None
        if self.kind == "emu":
            return b.ctypes.data_as(C.c_void_p)
        return C.c_void_p(b.data_ptr())

    def call(self, name, *args):
        rc = getattr(self.lib, name)(*args)
        if rc != 0:
            raise RuntimeError("%s rc=%d: %s" % (name, rc, self.lib.mn_last_error().decode()))

    # ---- convenience wrappers over the ABI
    def geom(self, x_shape, w_shape, stride=1, padding=0, dilation=1, groups=1):
        N, Cc, H, W = x_shape
        O, Cg, KH, KW = w_shape
        p2 = lambda v: (v, v) if isinstance(v, int) else tuple(v)
        (sh, sw), (ph, pw), (dh, dw) = p2(stride), p2(padding), p2(dilation)
        return _lib.ConvGeom(N, Cc, H, W, O, KH, KW, sh, sw, ph, pw, dh, dw, groups)

    def actq(self, mode=0, bits=8, q_type=0, qp=None, flags=0):
        self._qp_keep = qp
        return _lib.ActQ(mode, bits, q_type, flags, self.ptr(qp).value if qp is not None else None)

    def wq(self, mode=0, bits=8, q_type=0, per_channel=0, scale=None):
        self._ws_keep = scale
        return _lib.WQ(mode, bits, q_type, per_channel, self.ptr(scale).value if scale is not None else None)

    def conv_fwd(self, g, aq, x, w, b, algo, wq=None):
        Ho = (g.H + 2 * g.pad_h - g.dil_h * (g.KH - 1) - 1) // g.stride_h + 1
        Wo = (g.W + 2 * g.pad_w - g.dil_w * (g.KW - 1) - 1) // g.stride_w + 1
        y = self.empty((g.N, g.O, Ho, Wo))
        nb = self.lib.mn_conv2d_ws_bytes(C.byref(g), 0, algo)
        ws = self.empty(max(4, nb // 4 + 4))
        self.call("mn_conv2d_fwd", C.byref(g), C.byref(aq), C.byref(wq) if wq is not None else None, self.ptr(x), self.ptr(w), self.ptr(b), self.ptr(y), self.ptr(ws), nb, algo, self.stream)
        return y

    def conv_bwd_data(self, g, aq, gy, w, x, algo, wq=None):
        dx = self.empty((g.N, g.C, g.H, g.W))
        nb = self.lib.mn_conv2d_ws_bytes(C.byref(g), 1, algo)
        ws = self.empty(max(4, nb // 4 + 4))
        self.call("mn_conv2d_bwd_data", C.byref(g), C.byref(aq), C.byref(wq) if wq is not None else None, self.ptr(gy), self.ptr(w), self.ptr(x), self.ptr(dx), self.ptr(ws), nb, algo, self.stream)
        return dx

    def conv_bwd_weight(self, g, aq, gy, x, algo, bias=True):
        dw = self.empty((g.O, g.C // g.groups, g.KH, g.KW))
        db = self.empty(g.O) if bias else None
        nb = self.lib.mn_conv2d_ws_bytes(C.byref(g), 2, algo)
        ws = self.empty(max(4, nb // 4 + 4))
        self.call("mn_conv2d_bwd_weight", C.byref(g), C.byref(aq), self.ptr(gy), self.ptr(x), self.ptr(dw), self.ptr(db), self.ptr(ws), nb, algo, self.stream)
        return dw, db
