"""K0 for one or several class heads in one launch (csrc/shared_conv_f16.hip -> shasta_shared_conv_multi_f32).

The reference runs one `Shasta` model per tracking class (tools/nusc_shasta/eval.py:86-101, official_val.sh): each holds its own
`shared_conv` (det3d/models/tracker/shasta.py:42-47) and convolves the SAME neck output (:223-228).  `SharedConvBank` keeps the
fp16 piece images of the heads' `shared_conv.0 / .1` tensors side by side and produces every head's `example['bev_feature']` from
one read of the map.  Arithmetic: fp32 in, fp32 out, fp32 accumulation, products from two fp16 pieces per operand (the "f16x2"
arithmetic of `Shasta.arithmetic`); eval-mode BatchNorm only.  No CPU path.
"""
import ctypes as C
import weakref

import torch

from . import hip


def _tensors(m):
    conv, bn = m.shared_conv[0], m.shared_conv[1]
    return [conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var]


class SharedConvBank:
    """Packed fp16 images of the shared_conv of `models` (1 to 8 Shasta modules with the same in_channels)."""

    MAX_HEADS = 8

    def __init__(self, models):
        models = list(models)
        if not 1 <= len(models) <= self.MAX_HEADS:
            raise ValueError("SharedConvBank takes 1 to %d models" % self.MAX_HEADS)
        cin = {m.shared_conv[0].in_channels for m in models}
        if len(cin) != 1 or any(m.shared_conv[0].out_channels != 64 for m in models):
            raise ValueError("SharedConvBank: every head must be Conv2d(Cin -> 64) with the same Cin")
        # weak references: a model that keeps its own one-head bank (Shasta._conv_bank) must not form a cycle with it - the bank holds
        # device buffers that `del model` should free at once, without a pass of the cyclic collector
        self._models = [weakref.ref(m) for m in models]
        self.in_channels = cin.pop()
        self.cin_padded = (self.in_channels + 15) // 16 * 16
        self._packed = None
        self._key = None
        self._ws = None

    @property
    def models(self):
        ms = [r() for r in self._models]
        if any(m is None for m in ms):
            raise hip.ShastaHipError("SharedConvBank: a model of this bank no longer exists")
        return ms

    def supported(self, H, W):
        return bool(hip.load().shasta_shared_conv_f16x2_supported(self.cin_padded, H, W))

    def _ensure_packed(self, dev):
        key = tuple((t.data_ptr(), t._version) for m in self.models for t in _tensors(m)) + (str(dev),)
        if self._packed is not None and self._key == key:
            return
        lib = hip.load()
        stride = lib.shasta_shared_conv_f16x2_packed_bytes(self.cin_padded)
        self._stride = (stride + 255) // 256 * 256
        self._packed = torch.empty(len(self.models) * self._stride, dtype=torch.uint8, device=dev)
        for i, m in enumerate(self.models):
            ts = [t.detach() for t in _tensors(m)]
            if any((not t.is_cuda) or t.dtype != torch.float32 for t in ts):
                raise hip.ShastaHipError("shared_conv parameters must be fp32 device tensors (call .cuda() first; there is no CPU path)")
            w = ts[0].contiguous()
            if self.cin_padded != self.in_channels:  # zero channels add exactly
                wp = torch.zeros(64, self.cin_padded, 3, 3, device=dev)
                wp[:, :self.in_channels] = w
                w = wp
            rest = [t.contiguous() for t in ts[1:]]  # kept alive until the launch (a copy freed inside the call would lend its block to the next one)
            hip.check(lib.shasta_shared_conv_pack_f16x2(hip.ptr(w), *[hip.ptr(t) for t in rest], float(m.shared_conv[1].eps),
                                                        self.cin_padded, C.c_void_p(self._packed.data_ptr() + i * self._stride), stride,
                                                        hip.stream_ptr()), "shasta_shared_conv_pack_f16x2")
        self._key = key

    def __call__(self, bev_map, prev_bev_map=None, bound=None):
        """(B, Cin, H, W) fp32 device map(s) -> list over heads of (B, H, W, 64) NHWC tensors; with prev_bev_map a pair of lists.
        bound: max |x| of the maps as the producer knows it (need not be tight, must hold within a factor 4: beyond that the result is
        NaN / Inf) - the pass that finds the maxima is then skipped (shasta_shared_conv_multi_bounded_f32)."""
        maps = [t for t in (bev_map, prev_bev_map) if t is not None]
        if not all(t.is_cuda for t in maps):
            raise hip.ShastaHipError("SharedConvBank needs device tensors; there is no CPU path")
        if any(m.training for m in self.models):
            raise hip.ShastaHipError("SharedConvBank is the inference operator (eval-mode BatchNorm): call .eval() on the models")
        lib = hip.load()
        dev = bev_map.device
        self._ensure_packed(dev)
        x = self._prep(bev_map)
        xp = None if prev_bev_map is None else self._prep(prev_bev_map)
        if xp is not None and xp.shape != x.shape:
            raise ValueError("bev_map and prev_bev_map must have the same shape")
        B, _, H, W = x.shape
        nh = len(self._models)
        if not lib.shasta_shared_conv_f16x2_supported(self.cin_padded, H, W):
            raise hip.ShastaHipError("shared_conv (fp16 form): map %dx%d not served by this kernel" % (H, W))
        outs = [torch.empty(B, H, W, 64, device=dev) for _ in range(nh)]
        outs_p = None if xp is None else [torch.empty(B, H, W, 64, device=dev) for _ in range(nh)]
        # (from three heads on: room for the piece image of the maps, cut once for all heads - 68 MB per 512 x 180 x 180 map)
        wsb = lib.shasta_shared_conv_multi_workspace_bytes_for(max(B, 1), self.cin_padded, H, W, nh, int(xp is not None))
        if self._ws is None or self._ws.numel() * 4 < wsb or self._ws.device != dev:
            self._ws = None
            self._ws = torch.empty((wsb + 3) // 4, dtype=torch.int32, device=dev)
        arr = (C.c_void_p * nh)(*[t.data_ptr() for t in outs])
        arr_p = None if outs_p is None else (C.c_void_p * nh)(*[t.data_ptr() for t in outs_p])
        if bound is not None:
            hip.check(lib.shasta_shared_conv_multi_bounded_f32(hip.ptr(x), hip.ptr(xp), B, self.cin_padded, H, W, C.c_void_p(self._packed.data_ptr()),
                                                               self._stride, nh, arr, arr_p, hip.ptr(self._ws), self._ws.numel() * 4, float(bound),
                                                               hip.stream_ptr()), "shasta_shared_conv_multi_bounded_f32")
        else:
            hip.check(lib.shasta_shared_conv_multi_f32(hip.ptr(x), hip.ptr(xp), B, self.cin_padded, H, W, C.c_void_p(self._packed.data_ptr()),
                                                       self._stride, nh, arr, arr_p, hip.ptr(self._ws), self._ws.numel() * 4, hip.stream_ptr()),
                      "shasta_shared_conv_multi_f32")
        return outs if outs_p is None else (outs, outs_p)

    def _prep(self, t):
        x = t.float().contiguous()
        if self.cin_padded != self.in_channels:
            xp = torch.zeros(x.shape[0], self.cin_padded, x.shape[2], x.shape[3], device=x.device)
            xp[:, :self.in_channels] = x
            x = xp
        return x
