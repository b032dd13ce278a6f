"""The drafts' U-Net VAEs (vae/practice/vae_original.py:77-181 class VAEUnet256, :184-403 class VAEUnet1024) over the
C-ABI kernels.

Encoder: three GroupNorm double-conv blocks whose UNPOOLED outputs stay as skip tensors + MaxPool2d(2), a middle block,
fc_mu / fc_logvar; decoder: tanh(fc_up), three times ConvTranspose2d(k2, s2) -> cat([up, skip]) -> double conv; a 1x1
head that returns logits.  Same constructor, attribute / parameter names, state_dict and return value
``(logits, mu, logvar)`` as the reference class; NCHW fp32 at the boundary, NHWC bf16 between the kernels.

Each layer is one autograd node whose forward and backward are library calls (conv3x3 / ConvTranspose / linear implicit
GEMMs, GroupNorm + ReLU + dropout, max-pool, reparameterisation, 1x1 head).  torch holds the tensors, records the graph,
concatenates the skip tensors and repacks the fp32 master weights into the kernels' bf16 operand layouts.  No CPU path.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .variants import DoubleConvGN, maxpool2_nhwc

EPI_BF16, EPI_TANH, EPI_F32, EPI_F32_SPLIT = 0, 1, 2, 3
WGRAD_PLAIN, WGRAD_CONVT = 0, 2


def _wgrad(P, Q, M, N, K, ldp, ldq, mode, H, W, taps, out, dims4, strides4, s):
    """out (parameter layout) = fold of the split-K slabs of dW[m][tap][n] = sum_k P[k][m] Q[k(+tap)][n].
    dims4: the slab's own dims (walked in order, coalesced); strides4: where each of them lands in ``out``."""
    ns = _lib.call("vae_wgrad_ksplit", M, N, taps, K)
    slab = torch.empty(ns, M, taps, N, device=P.device)
    _lib.call("vae_wgrad_gemm", P.data_ptr(), Q.data_ptr(), slab.data_ptr(), M, N, K, ldp, ldq, mode, H, W, ns, s)
    _lib.call("vae_reduce_slabs_scatter", slab.data_ptr(), ns, M * taps * N, 0, out.data_ptr(), _lib.long4(*dims4),
              _lib.long4(*strides4), None, 0, 0, s)
    return out


def _colsum(x2d, s):
    rows, C = x2d.shape
    nb = _lib.call("vae_colsum_blocks", rows, C)
    partial = torch.empty(nb, C, device=x2d.device)
    out = torch.empty(C, device=x2d.device)
    _lib.call("vae_colsum_bf16", x2d.data_ptr(), rows, C, partial.data_ptr(), out.data_ptr(), s)
    return out


class _ConvT2x2(torch.autograd.Function):
    """nn.ConvTranspose2d(Ci, Co, 2, stride=2) on NHWC bf16 (vae_original.py:110,113,116,162-169).  No normalisation
    follows it here, so its bias has a real gradient: the column sums of the output gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        n, H, W, Ci = x.shape
        Co = weight.shape[1]
        s = _lib.current_stream()
        x = x.contiguous()
        wp = weight.detach().permute(2, 3, 1, 0).contiguous().to(torch.bfloat16)        # [(kh,kw,co)][ci]
        b = bias.detach().float().contiguous()
        y = torch.empty(n, 2 * H, 2 * W, Co, device=x.device, dtype=torch.bfloat16)
        _lib.call("vae_convt2x2_fwd_gemm", x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, n, H, W, Ci, Co, s)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        n, H, W, Ci = x.shape
        Co = weight.shape[1]
        s = _lib.current_stream()
        dy = dy.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            wd = weight.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)     # [ci][(kh,kw,co)]
            dx = torch.empty_like(x)
            _lib.call("vae_convt2x2_dgrad_gemm", dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), n, H, W, Ci, Co, s)
        dw = torch.empty(Ci, Co, 2, 2, device=x.device)
        # slab [co][tap][ci] -> parameter [ci][co][tap]
        _wgrad(dy, x, Co, Ci, n * H * W, Co, Ci, WGRAD_CONVT, H, W, 4, dw, (Co, 4, Ci, 1), (4, 1, 4 * Co, 0), s)
        db = _colsum(dy.view(-1, Co), s)
        return dx, dw, db


class _LatentHeads(torch.autograd.Function):
    """fc_mu and fc_logvar (vae_original.py:102-103,155-156) as ONE GEMM over the concatenated [2L][F] weight.  a is the
    middle block's output flattened in NHWC order; the weights' columns are permuted from the reference's NCHW flatten
    (index c*HW + p) to that order when packed, and the weight gradient is folded straight back into NCHW order."""

    @staticmethod
    def forward(ctx, a, w_mu, b_mu, w_lv, b_lv, hw, C):
        M, Fdim = a.shape
        L = w_mu.shape[0]
        s = _lib.current_stream()
        a = a.contiguous()
        wcat = torch.cat([w_mu.detach(), w_lv.detach()], 0).view(2 * L, C, hw).permute(0, 2, 1).reshape(2 * L, Fdim)
        wcat = wcat.contiguous().to(torch.bfloat16)
        bcat = torch.cat([b_mu.detach(), b_lv.detach()]).float().contiguous()
        mu = torch.empty(M, L, device=a.device)
        lv = torch.empty(M, L, device=a.device)
        _lib.call("vae_linear_gemm", a.data_ptr(), wcat.data_ptr(), bcat.data_ptr(), mu.data_ptr(), lv.data_ptr(), None, None,
                  M, 2 * L, Fdim, EPI_F32_SPLIT, s)
        ctx.save_for_backward(a, wcat)
        ctx.geom = (hw, C, L)
        return mu, lv

    @staticmethod
    def backward(ctx, dmu, dlv):
        a, wcat = ctx.saved_tensors
        hw, C, L = ctx.geom
        M, Fdim = a.shape
        s = _lib.current_stream()
        dcat32 = torch.cat([dmu, dlv], 1).contiguous()
        dcat = dcat32.to(torch.bfloat16)
        wt = wcat.t().contiguous()                                                    # [F][2L]
        da = torch.empty(M, Fdim, device=a.device, dtype=torch.bfloat16)
        _lib.call("vae_linear_gemm", dcat.data_ptr(), wt.data_ptr(), None, da.data_ptr(), None, None, None, M, Fdim, 2 * L,
                  EPI_BF16, s)
        dw = torch.empty(2 * L, Fdim, device=a.device)
        # slab [row][f = p*C + c] -> parameter [row][c*hw + p]
        _wgrad(dcat, a, 2 * L, Fdim, M, 2 * L, Fdim, WGRAD_PLAIN, 1, 1, 1, dw, (2 * L, hw, C, 1), (Fdim, 1, hw, 0), s)
        db = dcat32.sum(0)
        return da, dw[:L], db[:L], dw[L:], db[L:], None, None


class _Reparam(torch.autograd.Function):
    """z = mu + exp(0.5 logvar) * eps (vae_original.py:178-181); eps injected or from the Philox eps stream."""

    @staticmethod
    def forward(ctx, mu, logvar, eps, rng):
        seed, step, rank = rng
        mu, logvar = mu.contiguous(), logvar.contiguous()
        ne = mu.numel()
        z = torch.empty(mu.shape, device=mu.device, dtype=torch.bfloat16)
        klp = torch.empty(_lib.call("vae_reparam_grid", ne), device=mu.device)
        _lib.call("vae_reparam_kl_fwd", mu.data_ptr(), logvar.data_ptr(), _lib.ptr(eps), None, z.data_ptr(), None,
                  klp.data_ptr(), ne, seed, step, rank, _lib.current_stream())
        ctx.save_for_backward(mu, logvar, eps)
        ctx.rng = rng
        return z

    @staticmethod
    def backward(ctx, dz):
        mu, logvar, eps = ctx.saved_tensors
        seed, step, rank = ctx.rng
        dz = dz.float().contiguous()
        dmu, dlv = torch.empty_like(mu), torch.empty_like(logvar)
        _lib.call("vae_reparam_kl_bwd", dz.data_ptr(), mu.data_ptr(), logvar.data_ptr(), _lib.ptr(eps), None, None, 0.0,
                  dmu.data_ptr(), dlv.data_ptr(), None, mu.numel(), mu.size(-1), seed, step, rank, _lib.current_stream())
        return dmu, dlv, None, None


class _FcUpTanh(torch.autograd.Function):
    """tanh(fc_up(z)) (vae_original.py:107,159-160), emitted in NHWC order: the weight's rows are permuted when packed."""

    @staticmethod
    def forward(ctx, z, w, b, hw, C):
        M, L = z.shape
        Fdim = w.shape[0]
        s = _lib.current_stream()
        z = z.contiguous()
        wp = w.detach().view(C, hw, L).permute(1, 0, 2).reshape(Fdim, L).contiguous().to(torch.bfloat16)
        bp = b.detach().float().view(C, hw).t().reshape(Fdim).contiguous()
        h = torch.empty(M, Fdim, device=z.device, dtype=torch.bfloat16)
        _lib.call("vae_linear_gemm", z.data_ptr(), wp.data_ptr(), bp.data_ptr(), h.data_ptr(), None, None, None, M, Fdim, L,
                  EPI_TANH, s)
        ctx.save_for_backward(z, wp, h)
        ctx.geom = (hw, C)
        return h

    @staticmethod
    def backward(ctx, dh):
        z, wp, h = ctx.saved_tensors
        hw, C = ctx.geom
        M, L = z.shape
        Fdim = wp.shape[0]
        s = _lib.current_stream()
        dh = dh.contiguous()
        dpre = torch.empty_like(h)
        nr = _lib.call("vae_tanh_bwd_rows", M)
        colp = torch.empty(nr, Fdim, device=z.device)
        _lib.call("vae_tanh_bwd", dh.data_ptr(), h.data_ptr(), dpre.data_ptr(), colp.data_ptr(), M, Fdim, s)
        db = colp.sum(0).view(hw, C).t().reshape(Fdim)
        wt = wp.t().contiguous()                                                      # [L][F]
        dz = torch.empty(M, L, device=z.device)
        _lib.call("vae_linear_gemm", dpre.data_ptr(), wt.data_ptr(), None, dz.data_ptr(), None, None, None, M, L, Fdim,
                  EPI_F32, s)
        dw = torch.empty(Fdim, L, device=z.device)
        # slab [f = p*C + c][l] -> parameter [c*hw + p][l]
        _wgrad(dpre, z, Fdim, L, M, Fdim, L, WGRAD_PLAIN, 1, 1, 1, dw, (hw, C, L, 1), (L, hw * L, 1, 0), s)
        return dz.to(torch.bfloat16), dw, db, None, None


class _Conv1x1Out(torch.autograd.Function):
    """nn.Conv2d(64, Cout, 1) -> logits, NCHW fp32 (vae_original.py:119,173)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        n, H, W, C = x.shape
        Co = weight.shape[0]
        x = x.contiguous()
        w = weight.detach().float().reshape(Co, C).contiguous()
        b = bias.detach().float().contiguous()
        out = torch.empty(n, Co, H, W, device=x.device)
        _lib.call("vae_conv1x1_out_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), n, H, W, Co,
                  _lib.current_stream())
        ctx.save_for_backward(x, w)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        n, H, W, C = x.shape
        Co = w.shape[0]
        g = g.float().contiguous()
        nb = _lib.call("vae_conv1x1_out_bwd_blocks", n, H, W)
        partial = torch.empty(nb, (Co + 1) * 64, device=x.device)
        sums = torch.empty((Co + 1) * 64, device=x.device)
        dx = torch.empty_like(x)
        _lib.call("vae_conv1x1_out_bwd", g.data_ptr(), x.data_ptr(), w.data_ptr(), partial.data_ptr(), sums.data_ptr(),
                  dx.data_ptr(), n, H, W, Co, _lib.current_stream())
        return dx, sums[:Co * 64].view(Co, C, 1, 1), sums[Co * 64:Co * 64 + Co]


class UpConvT2x2(nn.ConvTranspose2d):
    """``nn.ConvTranspose2d(Ci, Co, 2, stride=2)`` with the reference's parameter names; ``forward_nhwc`` runs the kernels."""

    def __init__(self, in_channels, out_channels):
        super().__init__(in_channels, out_channels, 2, stride=2)

    def forward_nhwc(self, x):
        if not x.is_cuda or x.dtype != torch.bfloat16:
            raise RuntimeError("UpConvT2x2 takes an NHWC bf16 tensor on the GPU (no CPU fallback)")
        _lib.load()
        return _ConvT2x2.apply(x, self.weight, self.bias)

    def forward(self, x):
        return self.forward_nhwc(x.permute(0, 2, 3, 1).to(torch.bfloat16)).permute(0, 3, 1, 2).float()


class _Clamp(torch.autograd.Function):
    """torch.clamp(x, lo, hi) on fp32 (vae_original.py:390); the gradient passes where lo <= x <= hi."""

    @staticmethod
    def forward(ctx, x, lo, hi):
        x = x.contiguous()
        out = torch.empty_like(x)
        _lib.call("vae_clamp_f32", x.data_ptr(), None, lo, hi, out.data_ptr(), x.numel(), _lib.current_stream())
        ctx.save_for_backward(x)
        ctx.lim = (lo, hi)
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g.float().contiguous()
        dx = torch.empty_like(x)
        _lib.call("vae_clamp_f32", x.data_ptr(), g.data_ptr(), ctx.lim[0], ctx.lim[1], dx.data_ptr(), x.numel(),
                  _lib.current_stream())
        return dx, None, None


class _UnetVAE(nn.Module):
    """The two U-Net VAEs of the draft script share this body; a subclass gives the channel plan.

    ENC: (name, in, out) of the encoder blocks, the last one being ``middle`` (a MaxPool2d(2) behind every other one);
    DEC: (up name, up in, up out, refine name, refine in, refine out); XAVIER: xavier_uniform_(gain=0.1) on the latent
    heads; LOGVAR_CLAMP: clamp applied to logvar before it is used and returned, or None."""

    ENC = ()
    DEC = ()
    XAVIER = False
    LOGVAR_CLAMP = None

    def __init__(self, in_channels, latent_dims, p_dropout, image_size):
        super().__init__()
        down = 1 << (len(self.ENC) - 1)
        if image_size % down or latent_dims % 64:
            raise ValueError(f"{type(self).__name__}: image_size must be a multiple of {down} and latent_dims a multiple of 64")
        self.image_size = image_size
        for i, (name, ci, co) in enumerate(self.ENC):
            setattr(self, name, DoubleConvGN(ci, co, p_dropout))          # conv0 takes 3 channels whatever in_channels says
            if i < len(self.ENC) - 1:
                setattr(self, f"pool{i}", nn.MaxPool2d(2))
        C, H, W = self._get_flattened_shape(in_channels, p_dropout)
        self.chw = (C, H, W)
        self.flattened_dim = C * H * W
        self.fc_mu = nn.Linear(self.flattened_dim, latent_dims)
        self.fc_logvar = nn.Linear(self.flattened_dim, latent_dims)
        if self.XAVIER:
            nn.init.xavier_uniform_(self.fc_mu.weight, gain=0.1)
            nn.init.xavier_uniform_(self.fc_logvar.weight, gain=0.1)
        self.fc_up = nn.Linear(latent_dims, C * W * H)
        for up, uci, uco, ref, rci, rco in self.DEC:
            setattr(self, up, UpConvT2x2(uci, uco))
            setattr(self, ref, DoubleConvGN(rci, rco, p_dropout))
        self.head = nn.Conv2d(64, 3, 1, padding=0, stride=1)
        # one dropout stream per block; stream 4 is the eps stream of the reparameterisation kernel
        for sid, name in enumerate(self.block_names()):
            getattr(self, name).stream_id = sid if sid < 4 else sid + 1
        self.rng_seed, self.rng_step, self.rank = 1337, 0, 0

    @classmethod
    def block_names(cls):
        return tuple(e[0] for e in cls.ENC) + tuple(d[3] for d in cls.DEC)

    def _get_flattened_shape(self, in_channels, p_dropout):
        """the reference pushes zeros(1, C, S, S) through the encoder in train mode (vae_original.py:122-138,343-362):
        the only lasting effect is one Bernoulli mask drawn per block from the global generator when p > 0 (GroupNorm
        keeps no running state) -- drawn here on the same shapes so that later initialisations see the same generator
        state."""
        s = self.image_size
        for i, (_, _, co) in enumerate(self.ENC):
            if p_dropout > 0:
                F.dropout(torch.zeros(1, co, s, s), p_dropout, True)
            if i < len(self.ENC) - 1:
                s //= 2
        return self.ENC[-1][2], s, s

    def forward(self, x, eps=None, keep_masks=None):
        if not x.is_cuda:
            raise RuntimeError(f"{type(self).__name__} runs on the GPU only (no CPU fallback)")
        _lib.load()
        km = {}
        for k, v in (keep_masks or {}).items():
            km[k] = v.permute(0, 2, 3, 1).contiguous().to(torch.uint8)
        for name in self.block_names():
            blk = getattr(self, name)
            blk.rng_seed, blk.rank = self.rng_seed, self.rank
        skips = []
        h = x
        for i, (name, _, _) in enumerate(self.ENC):
            h = getattr(self, name).forward_nhwc(h, keep_mask=km.get(name))
            if i < len(self.ENC) - 1:
                skips.append(h)
                h = maxpool2_nhwc(h)
        n, hh, ww, C = h.shape
        mu, logvar = _LatentHeads.apply(h.reshape(n, hh * ww * C), self.fc_mu.weight, self.fc_mu.bias,
                                        self.fc_logvar.weight, self.fc_logvar.bias, hh * ww, C)
        if self.LOGVAR_CLAMP is not None:
            logvar = _Clamp.apply(logvar, float(self.LOGVAR_CLAMP[0]), float(self.LOGVAR_CLAMP[1]))
        z = self.reparameterize(mu, logvar, eps)
        h = _FcUpTanh.apply(z, self.fc_up.weight, self.fc_up.bias, hh * ww, C).view(n, hh, ww, C)
        for up, _, _, ref, _, _ in self.DEC:
            u = getattr(self, up).forward_nhwc(h)
            h = getattr(self, ref).forward_nhwc(torch.cat([u, skips.pop()], dim=-1), keep_mask=km.get(ref))
        out = _Conv1x1Out.apply(h, self.head.weight, self.head.bias)
        return out, mu, logvar

    def reparameterize(self, mu, logvar, eps=None):
        """reference signature (vae_original.py:178-181) -> z (bf16, the decoder's operand type)."""
        step = self.rng_step
        if eps is None:
            self.rng_step += 1
        else:
            eps = eps.float().contiguous()
        return _Reparam.apply(mu, logvar, eps, (self.rng_seed, step, self.rank))


class VAEUnet256(_UnetVAE):
    """Drop-in for the drafts' ``VAEUnet256(in_channels, latent_dims, p_dropout, image_size)``
    (vae/practice/vae_original.py:77-181): same constructor (consuming torch's RNG in the same order, so the same seed
    gives the same initial weights), same state_dict, ``forward(x) -> (logits, mu, logvar)``.
    ``eps`` / ``keep_masks`` (dict block name -> uint8 NCHW keep flags) replace the Philox draws for parity tests."""

    ENC = (("conv0", 3, 64), ("conv1", 64, 128), ("conv2", 128, 256), ("middle", 256, 256))
    DEC = (("up2", 256, 128, "refine2", 128 + 256, 128), ("up1", 128, 64, "refine1", 64 + 128, 64),
           ("up0", 64, 64, "refine0", 64 + 64, 64))
    XAVIER = True


class VAEUnet1024(_UnetVAE):
    """Drop-in for the drafts' ``VAEUnet1024`` (vae/practice/vae_original.py:184-403; of its two ``__init__`` bodies the
    second, :295-340, is the one Python keeps): five levels up to 1024 channels, default initialisation of the latent
    heads, and logvar clamped to [-5, 5] before use and return (:390; the [-10, 10] clamp inside its ``reparameterize``,
    :365, then never binds)."""

    ENC = (("conv0", 3, 64), ("conv1", 64, 128), ("conv2", 128, 256), ("conv3", 256, 512), ("middle", 512, 1024))
    DEC = (("up3", 1024, 512, "refine3", 1024, 512), ("up2", 512, 256, "refine2", 512, 256),
           ("up1", 256, 128, "refine1", 256, 128), ("up0", 128, 64, "refine0", 128, 64))
    LOGVAR_CLAMP = (-5.0, 5.0)


def vae_loss_mse(x, x_hat, mu, logvar, kl_weight):
    """``vae_loss_mse`` of the draft script (vae/practice/vae_original.py:527-532): (sum (x_hat - x)^2 + kl_weight * KL) / N
    on the head's logits, through the fused loss kernels."""
    from .train import Config, vae_loss
    return vae_loss(x_hat, x, mu, logvar, "mse", None, Config(kl_weight=float(kl_weight), percep_weight=0.0))


def vae_loss_bce(x, x_hat, mu, logvar):
    """``vae_loss_bce`` of the draft script (vae/practice/vae_original.py:513-518):
    mean over all elements of BCE-with-logits on logits clamped to [-10, 10], plus the KL SUM with weight 1 (no division
    by the batch) -- the fused loss kernels with the element count as the divisor and as the KL weight."""
    from .train import RECON_KINDS, _ElboLoss
    if not x_hat.is_cuda:
        raise RuntimeError("vae_loss_bce runs on the GPU only (no CPU fallback)")
    n = x_hat.numel()
    return _ElboLoss.apply(x_hat, x, mu, logvar, RECON_KINDS["bce_logits"], float(n), float(n))
