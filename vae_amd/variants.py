"""The GroupNorm form of the double-conv block from the reference's draft models
(vae/practice/vae_conv.py:20-36, vae/practice/vae_original.py:58-74): conv3x3 -> GroupNorm(8) -> ReLU, twice,
then Dropout; the drafts' encoders put a MaxPool2d(2) behind it (vae_conv.py:50-59), which the block can fuse.

Same kernels as the BatchNorm model: the conv3x3 implicit GEMMs (forward, data gradient, weight gradient) and
the GroupNorm + ReLU + dropout + pool kernels, all through the C-ABI library; torch only holds the tensors,
records the autograd graph and converts the NCHW fp32 boundary.  No CPU path: the ops raise without the library.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib

GN_GROUPS, GN_EPS = 8, 1e-5


def _conv_fwd(x, weight, bias, n, H, W, Ci, Co, s):
    """x: NHWC bf16 [n,H,W,Ci] with Ci % 64 == 0, or the 8-channel padded form when Ci <= 8."""
    y = torch.empty(n, H, W, Co, device=x.device, dtype=torch.bfloat16)
    b = bias.detach().float().contiguous()
    if Ci <= 8:
        wp = torch.zeros(Co, 16, 8, device=x.device, dtype=torch.bfloat16)
        w = weight.detach().float().contiguous()
        _lib.call("vae_pack_bf16", w.data_ptr(), wp.data_ptr(), _lib.long4(Co, Ci, 9, 1), _lib.long4(128, 1, 8, 0), 0, s)
        _lib.call("vae_conv3x3_c8_gemm", x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, n, H, W, Co, s)
    else:
        wp = weight.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)          # [Co][tap][Ci]
        _lib.call("vae_conv3x3_gemm", x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, n, H, W, Ci, Co, s)
    return y


def _conv_wgrad(dy, x, n, H, W, Ci, Co, s):
    dw = torch.empty(Co, Ci, 3, 3, device=dy.device, dtype=torch.float32)
    M = n * H * W
    if Ci <= 8:
        ns = _lib.call("vae_wgrad_ksplit", Co, 64, 2, M)
        slab = torch.empty(ns, Co, 2, 64, device=dy.device)
        _lib.call("vae_wgrad_gemm", dy.data_ptr(), x.data_ptr(), slab.data_ptr(), Co, 64, M, Co, 8, 3, H, W, ns, s)
        # slab [co][16 tap slots][8 channel slots] -> weight [co][ci][tap]; the padding slots are skipped
        _lib.call("vae_reduce_slabs_scatter", slab.data_ptr(), ns, Co * 128, 0, dw.data_ptr(), _lib.long4(Co, 16, 8, 1),
                  _lib.long4(9 * Ci, 1, 9, 0), _lib.long4(Co, 9, Ci, 1), 0, 0, s)
    elif _lib.call("vae_conv3_wgrad_patch_ok", H, W, Ci, Co) == 1:
        ns = _lib.call("vae_conv3_wgrad_patch_ksplit", n, H, W, Ci, Co)
        slab = torch.empty(ns, Co, 9, Ci, device=dy.device)
        _lib.call("vae_conv3_wgrad_patch", dy.data_ptr(), x.data_ptr(), slab.data_ptr(), n, H, W, Ci, Co, ns, s)
        # walk the slab [co][tap][ci] in its own order (coalesced reads), scatter into the weight's [co][ci][tap]
        _lib.call("vae_reduce_slabs_scatter", slab.data_ptr(), ns, 9 * Co * Ci, 0, dw.data_ptr(), _lib.long4(Co, 9, Ci, 1),
                  _lib.long4(9 * Ci, 1, 9, 0), None, 0, 0, s)
    else:
        ns = _lib.call("vae_wgrad_ksplit", Co, Ci, 9, M)
        slab = torch.empty(ns, Co, 9, Ci, device=dy.device)
        _lib.call("vae_wgrad_gemm", dy.data_ptr(), x.data_ptr(), slab.data_ptr(), Co, Ci, M, Co, Ci, 1, H, W, ns, s)
        # walk the slab [co][tap][ci] in its own order (coalesced reads), scatter into the weight's [co][ci][tap]
        _lib.call("vae_reduce_slabs_scatter", slab.data_ptr(), ns, 9 * Co * Ci, 0, dw.data_ptr(), _lib.long4(Co, 9, Ci, 1),
                  _lib.long4(9 * Ci, 1, 9, 0), None, 0, 0, s)
    return dw


class _ConvGnAct(torch.autograd.Function):
    """conv3x3(pad 1) -> GroupNorm(8) -> ReLU -> [dropout] -> [2x2 max-pool] as one autograd node."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, geom, act):
        n, H, W, Ci, Co = geom
        pool, p, mask, seed, step, stream_id, rank = act
        s = _lib.current_stream()
        y = _conv_fwd(x, weight, bias, n, H, W, Ci, Co, s)
        out = torch.empty((n, H // 2, W // 2, Co) if pool else (n, H, W, Co), device=x.device, dtype=torch.bfloat16)
        mean_rstd = torch.empty(n, GN_GROUPS, 2, device=x.device)
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        _lib.call("vae_gn_act_fwd", y.data_ptr(), g32.data_ptr(), b32.data_ptr(), out.data_ptr(), mean_rstd.data_ptr(),
                  n, H, W, Co, GN_GROUPS, GN_EPS, int(pool), float(p), None if mask is None else mask.data_ptr(),
                  seed, step, stream_id, rank, s)
        ctx.save_for_backward(x, y, mean_rstd, weight, g32, b32)
        ctx.geom, ctx.act = geom, act
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, y, mean_rstd, weight, g32, b32 = ctx.saved_tensors
        n, H, W, Ci, Co = ctx.geom
        pool, p, mask, seed, step, stream_id, rank = ctx.act
        s = _lib.current_stream()
        g_out = g_out.contiguous()
        part = torch.empty(n, 3, Co, device=y.device)
        sums = torch.empty(3 * Co, device=y.device)
        dy = torch.empty_like(y)
        _lib.call("vae_gn_act_bwd", g_out.data_ptr(), y.data_ptr(), g32.data_ptr(), b32.data_ptr(), mean_rstd.data_ptr(),
                  part.data_ptr(), sums.data_ptr(), dy.data_ptr(), n, H, W, Co, GN_GROUPS, int(pool), float(p),
                  None if mask is None else mask.data_ptr(), seed, step, stream_id, rank, s)
        dw = _conv_wgrad(dy, x, n, H, W, Ci, Co, s)
        dx = None
        if ctx.needs_input_grad[0]:
            if Ci <= 8:
                raise RuntimeError("the 8-channel first conv takes the network input and produces no input gradient")
            wd = weight.detach().flip(2, 3).permute(1, 2, 3, 0).contiguous().to(torch.bfloat16)     # [Ci][flipped tap][Co]
            dx = torch.empty(n, H, W, Ci, device=y.device, dtype=torch.bfloat16)
            _lib.call("vae_conv3x3_gemm", dy.data_ptr(), wd.data_ptr(), None, dx.data_ptr(), None, None, n, H, W, Co, Ci, s)
        return dx, dw, sums[2 * Co:], sums[:Co], sums[Co:2 * Co], None, None


class DoubleConvGN(nn.Module):
    """Drop-in for the drafts' ``DoubleConv(in_channels, out_channels, p_dropout)`` (vae/practice/vae_conv.py:20-36):
    same constructor, same parameter names (``double_conv.{0,1,3,4}.{weight,bias}``), NCHW fp32 in and out.
    ``forward_nhwc`` chains blocks without leaving the kernels' layout (NHWC bf16) and can fuse the encoder's
    MaxPool2d(2).  Dropout draws come from Philox(rng_seed, rng_step, stream_id, rank); ``rng_step`` advances
    with every training forward."""

    def __init__(self, in_channels, out_channels, p_dropout):
        super().__init__()
        if not (in_channels <= 8 or in_channels % 64 == 0) or out_channels % 64 != 0:
            raise ValueError("DoubleConvGN: in_channels must be <= 8 or a multiple of 64, out_channels a multiple of 64")
        self.double_conv = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, 3, padding=1), nn.GroupNorm(GN_GROUPS, out_channels), nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, 3, padding=1), nn.GroupNorm(GN_GROUPS, out_channels), nn.ReLU(inplace=True),
            nn.Dropout(p=p_dropout))
        self.in_channels, self.out_channels, self.p = in_channels, out_channels, float(p_dropout)
        self.rng_seed, self.rng_step, self.stream_id, self.rank = 1337, 0, 0, 0

    def forward_nhwc(self, x, pool=False, keep_mask=None):
        """x: NHWC bf16 [n,H,W,Cin] (Cin a multiple of 64), or NCHW fp32 when Cin <= 8 (the network input).
        keep_mask: optional uint8 [n,H,W,Cout] dropout keep flags (instead of the Philox stream)."""
        if not x.is_cuda:
            raise RuntimeError("DoubleConvGN runs on the GPU only (no CPU fallback)")
        _lib.load()
        sq = self.double_conv
        Ci, Co = self.in_channels, self.out_channels
        s = _lib.current_stream()
        if Ci <= 8:
            n, _, H, W = x.shape
            x8 = torch.empty(n, H, W, 8, device=x.device, dtype=torch.bfloat16)
            xc = x.detach().float().contiguous()
            _lib.call("vae_nchw_to_nhwc8", xc.data_ptr(), x8.data_ptr(), n, Ci, H, W, s)
            x = x8
        else:
            n, H, W, _ = x.shape
            x = x.contiguous()
        p = self.p if self.training else 0.0
        if pool and (H % 2 or W % 2):
            raise ValueError("pooling needs even H, W")
        step = self.rng_step
        if self.training and p > 0:
            self.rng_step += 1
        no_act = (False, 0.0, None, 0, 0, 0, 0)
        h = _ConvGnAct.apply(x, sq[0].weight, sq[0].bias, sq[1].weight, sq[1].bias, (n, H, W, Ci, Co), no_act)
        act = (bool(pool), p, keep_mask, self.rng_seed, step, self.stream_id, self.rank)
        return _ConvGnAct.apply(h, sq[3].weight, sq[3].bias, sq[4].weight, sq[4].bias, (n, H, W, Co, Co), act)

    def forward(self, x, keep_mask=None):
        """reference signature: NCHW fp32 in, NCHW fp32 out (no pooling)."""
        if self.in_channels > 8:
            x = x.permute(0, 2, 3, 1).to(torch.bfloat16)
        out = self.forward_nhwc(x, False, keep_mask)
        return out.permute(0, 3, 1, 2).float()


class _MaxPool2(torch.autograd.Function):
    """nn.MaxPool2d(2) on an NHWC bf16 tensor; the backward reads the route bytes the forward left."""

    @staticmethod
    def forward(ctx, x):
        n, H, W, C = x.shape
        x = x.contiguous()
        out = torch.empty(n, H // 2, W // 2, C, device=x.device, dtype=torch.bfloat16)
        route = torch.empty(n, H // 2, W // 2, C, device=x.device, dtype=torch.uint8)
        _lib.call("vae_maxpool2_fwd", x.data_ptr(), out.data_ptr(), route.data_ptr(), n, H, W, C, _lib.current_stream())
        ctx.save_for_backward(route)
        ctx.shape = (n, H, W, C)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (route,) = ctx.saved_tensors
        n, H, W, C = ctx.shape
        g_out = g_out.contiguous()
        dx = torch.empty(n, H, W, C, device=g_out.device, dtype=torch.bfloat16)
        _lib.call("vae_maxpool2_bwd", g_out.data_ptr(), None, route.data_ptr(), dx.data_ptr(), n, H, W, C, _lib.current_stream())
        return dx


def maxpool2_nhwc(x):
    """``nn.MaxPool2d(2)`` for the kernels' layout (NHWC bf16, even H and W, channels a multiple of 8): what the drafts'
    U-Net encoders apply to a block's output while keeping the unpooled tensor for the skip connection
    (vae/practice/vae_original.py:142-149)."""
    if not x.is_cuda or x.dtype != torch.bfloat16:
        raise RuntimeError("maxpool2_nhwc takes an NHWC bf16 tensor on the GPU (no CPU fallback)")
    _lib.load()
    return _MaxPool2.apply(x)
