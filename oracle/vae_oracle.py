"""CPU oracle for the conv-VAE training hot path of jbthejedi/vae.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (vae_amd/) may import
this file; it is used by tests/, by __graft_entry__.smoke() and by the
cpu_baseline leg of bench.py as the checker / the timed CPU baseline.

It restates, in plain functional fp32 PyTorch on the CPU, exactly what the
reference computes on the path

    VAEConv.forward  -> (recons, mu, logvar)        vae/models/vae.py:101-105
    vae_loss                                        vae/training/train.py:44-51
    train-step body                                 vae/training/train.py:327-339

Parity is pinned by tests/golden/*.npz, which oracle/make_fixtures.py generated
by importing the reference classes themselves in the build container (the
reference has no tests or golden vectors of its own).

The counter-based RNG (Philox4x32-10 and the integer/float transforms built on
it) is restated here in numpy; it is pinned by the Random123 known-answer
vectors (tests/test_oracle_philox.py).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nn.BatchNorm2d default, vae/models/vae.py:130
BN_MOMENTUM = 0.1      # nn.BatchNorm2d default

ENC_CHANNELS = (64, 128, 256, 256)   # vae/models/vae.py:34-44


# ----------------------------------------------------------------------------
# Parameter inventory (vae/models/vae.py:23-86).  Order == state_dict order.
# ----------------------------------------------------------------------------
def state_shapes(in_channels=3, out_channels=3, latent_dims=128, image_size=32):
    """Ordered {name: shape} of the 85-entry state_dict of VAEConv."""
    s = OrderedDict()
    ci = in_channels
    for i, co in enumerate(ENC_CHANNELS, start=1):
        for j, cin in ((0, ci), (3, co)):
            s[f"conv{i}.net.{j}.weight"] = (co, cin, 3, 3)
            s[f"conv{i}.net.{j}.bias"] = (co,)
            b = j + 1
            s[f"conv{i}.net.{b}.weight"] = (co,)
            s[f"conv{i}.net.{b}.bias"] = (co,)
            s[f"conv{i}.net.{b}.running_mean"] = (co,)
            s[f"conv{i}.net.{b}.running_var"] = (co,)
            s[f"conv{i}.net.{b}.num_batches_tracked"] = ()
        ci = co
    hw = image_size // 8                      # three MaxPool2d(2), vae.py:35-41
    flat = 256 * hw * hw
    s["fc_mu.weight"] = (latent_dims, flat)
    s["fc_mu.bias"] = (latent_dims,)
    s["fc_logvar.weight"] = (latent_dims, flat)
    s["fc_logvar.bias"] = (latent_dims,)
    s["fc_up.weight"] = (flat, latent_dims)
    s["fc_up.bias"] = (flat,)
    for name, bn, cin, co in (("up3", "bn3", 256, 128), ("up2", "bn2", 128, 64),
                              ("up1", "bn1", 64, 64)):
        s[f"{name}.weight"] = (cin, co, 2, 2)  # ConvTranspose2d is [Cin, Cout, kh, kw]
        s[f"{name}.bias"] = (co,)
        s[f"{bn}.weight"] = (co,)
        s[f"{bn}.bias"] = (co,)
        s[f"{bn}.running_mean"] = (co,)
        s[f"{bn}.running_var"] = (co,)
        s[f"{bn}.num_batches_tracked"] = ()
    s["head.weight"] = (out_channels, 64, 1, 1)
    s["head.bias"] = (out_channels,)
    return s


def is_param(name: str) -> bool:
    return not (name.endswith("running_mean") or name.endswith("running_var")
                or name.endswith("num_batches_tracked"))


def _uniform(shape, bound):
    return torch.empty(shape, dtype=torch.float32).uniform_(-bound, bound)


def construct_state(in_channels=3, out_channels=3, latent_dims=128, p_dropout=0.1,
                    image_size=32):
    """Restates VAEConv.__init__ (vae/models/vae.py:23-86) including its side
    effects, consuming torch's global CPU generator in the same order, so that
    ``torch.manual_seed(s); construct_state(...)`` equals
    ``torch.manual_seed(s); VAEConv(...).state_dict()``.

    * Conv2d / Linear / ConvTranspose2d default init: weight and bias both
      U(-1/sqrt(fan_in), 1/sqrt(fan_in)), fan_in = weight.size(1) * kh * kw
      (for ConvTranspose2d that is Cout*kh*kw).
    * the train-mode dummy forward of _get_flattened_dims (vae.py:78-86) draws
      one Bernoulli(1-p) mask per DoubleConv (when p > 0) and moves every
      encoder BatchNorm's running stats one step (num_batches_tracked = 1,
      running_var = 0.9, running_mean = 0.1 * batch mean).
    * xavier_uniform_(gain=0.1) on fc_mu / fc_logvar weights (vae.py:60-61).
    """
    shapes = state_shapes(in_channels, out_channels, latent_dims, image_size)
    sd = OrderedDict()

    def conv_like(wname, bname):
        shp = shapes[wname]
        fan_in = shp[1] * (shp[2] * shp[3] if len(shp) == 4 else 1)
        bound = 1.0 / math.sqrt(fan_in)
        sd[wname] = _uniform(shp, bound)
        sd[bname] = _uniform(shapes[bname], bound)

    def bn(prefix, c):
        sd[f"{prefix}.weight"] = torch.ones(c)
        sd[f"{prefix}.bias"] = torch.zeros(c)
        sd[f"{prefix}.running_mean"] = torch.zeros(c)
        sd[f"{prefix}.running_var"] = torch.ones(c)
        sd[f"{prefix}.num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)

    for i, co in enumerate(ENC_CHANNELS, start=1):
        conv_like(f"conv{i}.net.0.weight", f"conv{i}.net.0.bias")
        bn(f"conv{i}.net.1", co)
        conv_like(f"conv{i}.net.3.weight", f"conv{i}.net.3.bias")
        bn(f"conv{i}.net.4", co)

    # dummy forward on zeros(1, C, S, S) in train mode (vae.py:78-86).  The batch
    # variance is ~0 there, so every BatchNorm multiplies the rounding noise of
    # (x - mean) by rsqrt(1e-5) = 316 and the deeper running means are set by that
    # noise; it is reproduced only by running the very same ATen ops in the same
    # order, which is what this block does (F.batch_norm / F.dropout, not the
    # explicit formulas used elsewhere in this file).
    with torch.no_grad():
        x = torch.zeros(1, in_channels, image_size, image_size)
        for i, co in enumerate(ENC_CHANNELS, start=1):
            for j in (0, 3):
                b = f"conv{i}.net.{j + 1}"
                x = F.conv2d(x, sd[f"conv{i}.net.{j}.weight"], sd[f"conv{i}.net.{j}.bias"], padding=1)
                x = F.batch_norm(x, sd[f"{b}.running_mean"], sd[f"{b}.running_var"],
                                 sd[f"{b}.weight"], sd[f"{b}.bias"], True, BN_MOMENTUM, BN_EPS)
                sd[f"{b}.num_batches_tracked"] = torch.tensor(1, dtype=torch.int64)
                x = torch.relu_(x)
            x = F.dropout(x, p_dropout, True)     # draws the Bernoulli mask when p > 0
            if i <= 3:
                x = F.max_pool2d(x, 2)

    for name in ("fc_mu", "fc_logvar"):
        conv_like(f"{name}.weight", f"{name}.bias")
    for name in ("fc_mu", "fc_logvar"):
        shp = shapes[f"{name}.weight"]
        a = 0.1 * math.sqrt(6.0 / (shp[0] + shp[1]))
        sd[f"{name}.weight"] = _uniform(shp, a)
    conv_like("fc_up.weight", "fc_up.bias")
    for name, b, co in (("up3", "bn3", 128), ("up2", "bn2", 64), ("up1", "bn1", 64)):
        conv_like(f"{name}.weight", f"{name}.bias")
        bn(b, co)
    conv_like("head.weight", "head.bias")
    # re-order to state_dict order
    return OrderedDict((k, sd[k]) for k in shapes)


# ----------------------------------------------------------------------------
# Layers
# ----------------------------------------------------------------------------
def batch_norm(x, gamma, beta, running_mean, running_var, train: bool,
               stats_out: Optional[dict] = None, prefix: str = ""):
    """nn.BatchNorm2d forward (vae/models/vae.py:130,134,65,68,71).

    train: normalise with the batch mean / biased variance over (N,H,W); the
    running stats move by momentum 0.1 with the UNBIASED variance.
    eval: normalise with the running stats.  New running stats are written to
    ``stats_out`` (the inputs are not modified).
    """
    if train:
        n = x.numel() // x.shape[1]
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if stats_out is not None:
            with torch.no_grad():
                unb = var * (n / max(n - 1, 1))
                stats_out[f"{prefix}.running_mean"] = (1 - BN_MOMENTUM) * running_mean + BN_MOMENTUM * mean
                stats_out[f"{prefix}.running_var"] = (1 - BN_MOMENTUM) * running_var + BN_MOMENTUM * unb
    else:
        mean, var = running_mean, running_var
    xhat = (x - mean[None, :, None, None]) * torch.rsqrt(var + BN_EPS)[None, :, None, None]
    return xhat * gamma[None, :, None, None] + beta[None, :, None, None]


def dropout(x, p: float, train: bool, mask: Optional[torch.Tensor]):
    """nn.Dropout (vae/models/vae.py:137): train -> x * mask / (1-p); eval -> x.
    ``mask`` is the injected keep-mask (1 = keep)."""
    if not train or p == 0.0:
        return x
    assert mask is not None, "train-mode dropout needs an injected keep-mask"
    return x * mask.to(x.dtype) / (1.0 - p)


def double_conv(x, sd, prefix, train, p_dropout, mask, stats_out=None):
    """DoubleConv.forward (vae/models/vae.py:125-141)."""
    for j in (0, 3):
        x = F.conv2d(x, sd[f"{prefix}.net.{j}.weight"], sd[f"{prefix}.net.{j}.bias"], padding=1)
        b = f"{prefix}.net.{j + 1}"
        x = batch_norm(x, sd[f"{b}.weight"], sd[f"{b}.bias"], sd[f"{b}.running_mean"],
                       sd[f"{b}.running_var"], train, stats_out, b)
        x = torch.relu(x)
    return dropout(x, p_dropout, train, mask)


GN_GROUPS = 8          # nn.GroupNorm(num_groups=8, ...), vae/practice/vae_conv.py:26,30
GN_EPS = 1e-5          # nn.GroupNorm default


def group_norm(x, gamma, beta, groups: int = GN_GROUPS, eps: float = GN_EPS):
    """nn.GroupNorm forward (vae/practice/vae_conv.py:26,30; vae/practice/vae_original.py:64,68): per image
    and group of C/groups channels, mean and BIASED variance over (C/groups, H, W); the same in train and
    eval mode (no running statistics); then the per-channel affine."""
    n, c, h, w = x.shape
    xg = x.reshape(n, groups, (c // groups) * h * w)
    mean = xg.mean(dim=2, keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=2, keepdim=True)
    xhat = ((xg - mean) * torch.rsqrt(var + eps)).reshape(n, c, h, w)
    return xhat * gamma[None, :, None, None] + beta[None, :, None, None]


def double_conv_gn(x, sd, prefix, train, p_dropout, mask):
    """The GroupNorm form of the double-conv block (vae/practice/vae_conv.py:20-36):
    conv3x3, GroupNorm(8), ReLU, conv3x3, GroupNorm(8), ReLU, Dropout.  State names follow the
    draft's nn.Sequential: ``{prefix}.double_conv.{0,1,3,4}.{weight,bias}``."""
    for j in (0, 3):
        x = F.conv2d(x, sd[f"{prefix}.double_conv.{j}.weight"], sd[f"{prefix}.double_conv.{j}.bias"], padding=1)
        x = torch.relu(group_norm(x, sd[f"{prefix}.double_conv.{j + 1}.weight"], sd[f"{prefix}.double_conv.{j + 1}.bias"]))
    return dropout(x, p_dropout, train, mask)


def encode(x, sd, train=True, p_dropout=0.0, masks=None, stats_out=None):
    """VAEConv.encode (vae/models/vae.py:89-98) -> (mu, logvar)."""
    masks = masks or [None] * 4
    for i in range(1, 5):
        x = double_conv(x, sd, f"conv{i}", train, p_dropout, masks[i - 1], stats_out)
        if i <= 3:
            x = F.max_pool2d(x, 2)
    x = torch.flatten(x, start_dim=1)
    mu = F.linear(x, sd["fc_mu.weight"], sd["fc_mu.bias"])
    logvar = F.linear(x, sd["fc_logvar.weight"], sd["fc_logvar.bias"])
    return mu, logvar


def reparameterize(mu, logvar, eps):
    """VAEConv.reparameterize (vae/models/vae.py:108-111) with eps injected."""
    return mu + torch.exp(0.5 * logvar) * eps


def decode(z, sd, train=True, stats_out=None):
    """VAEConv.decode (vae/models/vae.py:114-122)."""
    flat = sd["fc_up.weight"].shape[0]
    hw = int(round(math.sqrt(flat // 256)))
    x = torch.tanh(F.linear(z, sd["fc_up.weight"], sd["fc_up.bias"])).view(-1, 256, hw, hw)
    for up, bn in (("up3", "bn3"), ("up2", "bn2"), ("up1", "bn1")):
        x = F.conv_transpose2d(x, sd[f"{up}.weight"], sd[f"{up}.bias"], stride=2)
        x = batch_norm(x, sd[f"{bn}.weight"], sd[f"{bn}.bias"], sd[f"{bn}.running_mean"],
                       sd[f"{bn}.running_var"], train, stats_out, bn)
        x = torch.relu(x)
    return torch.sigmoid(F.conv2d(x, sd["head.weight"], sd["head.bias"]))


def forward(x, sd, eps, train=True, p_dropout=0.0, masks=None, stats_out=None):
    """VAEConv.forward (vae/models/vae.py:101-105) -> (recons, mu, logvar)."""
    mu, logvar = encode(x, sd, train, p_dropout, masks, stats_out)
    z = reparameterize(mu, logvar, eps)
    return decode(z, sd, train, stats_out), mu, logvar


# ----------------------------------------------------------------------------
# Loss (vae/training/train.py:44-55 and the draft variants)
# ----------------------------------------------------------------------------
RECON_KINDS = ("l1", "mse", "bce", "bce_logits", "bce_logits_raw")


def recon_loss_sum(recon, target, kind: str):
    """reduction='sum' reconstruction term.
    l1  : F.l1_loss                                   vae/training/train.py:55
    mse : F.mse_loss                                  vae/practice/vae_conv.py:160-165
    bce : F.binary_cross_entropy (log clamped >= -100) vae/bak/vae_conv.py:183-190
    bce_logits : F.binary_cross_entropy_with_logits on logits clamped to
          [-10, 10]                                   vae/practice/vae_conv.py:146-151
    bce_logits_raw : F.binary_cross_entropy_with_logits itself (no clamp), i.e. vae_loss called with
          loss=F.binary_cross_entropy_with_logits     vae/training/train.py:46
    """
    if kind == "l1":
        return (recon - target).abs().sum()
    if kind == "mse":
        return ((recon - target) ** 2).sum()
    if kind == "bce":
        lr = torch.clamp(torch.log(recon), min=-100.0)
        l1r = torch.clamp(torch.log(1.0 - recon), min=-100.0)
        return -(target * lr + (1.0 - target) * l1r).sum()
    if kind == "bce_logits":
        xl = recon.clamp(-10.0, 10.0)
        # max(x,0) - x*t + log(1 + exp(-|x|))
        return (xl.clamp(min=0) - xl * target + torch.log1p(torch.exp(-xl.abs()))).sum()
    if kind == "bce_logits_raw":
        return (recon.clamp(min=0) - recon * target + torch.log1p(torch.exp(-recon.abs()))).sum()
    raise ValueError(kind)


def kl_sum(mu, logvar):
    """-0.5 * sum(1 + logvar - mu^2 - exp(logvar))   vae/training/train.py:47"""
    return -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp())


def vae_loss(recon, target, mu, logvar, kind="l1", kl_weight=1.0):
    """(recon_sum + kl_weight * KL) / N  (percep_weight = 0)  vae/training/train.py:50"""
    return (recon_loss_sum(recon, target, kind) + kl_weight * kl_sum(mu, logvar)) / recon.size(0)


# ----------------------------------------------------------------------------
# Train step (vae/training/train.py:327-339)
# ----------------------------------------------------------------------------
def clip_grad_norm(grads: Dict[str, torch.Tensor], max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_ (train.py:336): global L2 norm over all
    grads, coefficient max_norm/(norm+1e-6) clamped to <= 1.  Returns the
    pre-clip norm; grads are scaled in place."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return total


def adamw_step(params, grads, exp_avg, exp_avg_sq, step: int, lr=1e-4, betas=(0.9, 0.999),
               eps=1e-8, weight_decay=0.01):
    """torch.optim.AdamW defaults (train.py:310), single step ``step`` (1-based)."""
    b1, b2 = betas
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for k, p in params.items():
        g = grads[k]
        p.mul_(1 - lr * weight_decay)
        exp_avg[k].mul_(b1).add_(g, alpha=1 - b1)
        exp_avg_sq[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (exp_avg_sq[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(exp_avg[k], denom, value=-(lr / bc1))


def loss_and_grads(x, sd, eps, kind="l1", kl_weight=1.0, train=True, p_dropout=0.0, masks=None):
    """forward + loss + backward.  Returns (loss, recon, mu, logvar, grads, new_stats)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if is_param(k)}
    full = dict(sd)
    full.update(params)
    stats = {}
    recon, mu, logvar = forward(x, full, eps, train, p_dropout, masks, stats)
    loss = vae_loss(recon, x, mu, logvar, kind, kl_weight)
    gl = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(params[k]))
             for k, g in zip(params.keys(), gl)}
    return loss.detach(), recon.detach(), mu.detach(), logvar.detach(), grads, stats


def train_step(x, sd, opt_state, eps, kind="l1", kl_weight=1.0, p_dropout=0.0, masks=None,
               lr=1e-4, max_norm=1.0):
    """One reference train step on state ``sd`` (updated in place).
    opt_state = {"step": int, "exp_avg": {...}, "exp_avg_sq": {...}}."""
    loss, recon, mu, logvar, grads, stats = loss_and_grads(x, sd, eps, kind, kl_weight, True,
                                                           p_dropout, masks)
    gnorm = clip_grad_norm(grads, max_norm)
    opt_state["step"] += 1
    params = {k: v for k, v in sd.items() if is_param(k)}
    with torch.no_grad():
        adamw_step(params, grads, opt_state["exp_avg"], opt_state["exp_avg_sq"], opt_state["step"], lr)
        for k, v in stats.items():
            sd[k] = v
        for k in sd:
            if k.endswith("num_batches_tracked"):
                sd[k] = sd[k] + 1
    return loss, gnorm


def new_opt_state(sd):
    return {"step": 0,
            "exp_avg": {k: torch.zeros_like(v) for k, v in sd.items() if is_param(k)},
            "exp_avg_sq": {k: torch.zeros_like(v) for k, v in sd.items() if is_param(k)}}


# ----------------------------------------------------------------------------
# Counter-based RNG: Philox4x32-10 (Salmon et al., SC'11; Random123 v1.14).
# The reference draws eps with torch.randn_like (vae.py:110) and dropout masks
# with nn.Dropout (vae.py:137) from torch's global generator; the MI355X build
# replaces those with explicit Philox streams whose definition is below.
# ----------------------------------------------------------------------------
PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint32(0x9E3779B9)
PHILOX_W1 = np.uint32(0xBB67AE85)

STREAM_DROPOUT = (0, 1, 2, 3)   # one stream per DoubleConv
STREAM_EPS = 4


def philox4x32_10(ctr: np.ndarray, key: np.ndarray) -> np.ndarray:
    """ctr: [..., 4] uint32, key: [..., 2] uint32 (broadcastable) -> [..., 4] uint32."""
    c = [ctr[..., i].astype(np.uint32) for i in range(4)]
    k0 = np.broadcast_to(key[..., 0].astype(np.uint32), c[0].shape).copy()
    k1 = np.broadcast_to(key[..., 1].astype(np.uint32), c[0].shape).copy()
    mask32 = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for r in range(10):
            p0 = PHILOX_M0 * c[0].astype(np.uint64)
            p1 = PHILOX_M1 * c[2].astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & mask32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & mask32).astype(np.uint32)
            c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
            k0 = (k0 + PHILOX_W0).astype(np.uint32)
            k1 = (k1 + PHILOX_W1).astype(np.uint32)
    return np.stack(c, axis=-1)


def philox_bits(n: int, seed: int, stream: int, step: int, rank: int = 0) -> np.ndarray:
    """The uint32 stream element i (0 <= i < n) of a (seed, stream, step, rank)
    draw: word i%4 of philox(ctr = {i/4, step, stream, rank}, key = {seed_lo, seed_hi})."""
    nb = (n + 3) // 4
    ctr = np.zeros((nb, 4), dtype=np.uint32)
    ctr[:, 0] = np.arange(nb, dtype=np.uint64).astype(np.uint32)
    ctr[:, 1] = np.uint32(step & 0xFFFFFFFF)
    ctr[:, 2] = np.uint32(stream)
    ctr[:, 3] = np.uint32(rank)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    return philox4x32_10(ctr, key[None, :]).reshape(-1)[:n]


def dropout_threshold(p: float) -> int:
    """keep iff the element's 16-bit draw >= ceil(p * 2^16)  (P(keep) = 1 - thr/2^16)."""
    return int(math.ceil(p * 65536.0))


def philox_keep_mask(n: int, p: float, seed: int, stream: int, step: int, rank: int = 0) -> np.ndarray:
    """Element i draws 16 bits: half (i & 1) of word (i >> 1) & 3 of Philox block i >> 3, i.e. the
    32-bit stream word i >> 1 split low half first (one Philox block serves 8 elements)."""
    words = philox_bits((n + 1) // 2, seed, stream, step, rank)
    halves = np.stack([words & np.uint32(0xFFFF), words >> np.uint32(16)], axis=-1).reshape(-1)[:n]
    return (halves >= np.uint32(dropout_threshold(p))).astype(np.uint8)


# fp32-only Box-Muller built from individually rounded +,*,/ and sqrt (no fma,
# no libm), so that the HIP kernel and this restatement agree bit for bit.
_F = np.float32
_LN2 = _F(0.69314718055994530942)
_SQRT2 = _F(1.41421356237309504880)
_HALF_PI = _F(1.57079632679489661923)
# atanh-series coefficients 2/(2k+1)
_LOG_C = [_F(2.0 / (2 * k + 1)) for k in range(0, 6)]
# sin(x), cos(x) Taylor coefficients on |x| <= pi/4
_SIN_C = [_F(-1.0 / 6), _F(1.0 / 120), _F(-1.0 / 5040), _F(1.0 / 362880)]
_COS_C = [_F(-0.5), _F(1.0 / 24), _F(-1.0 / 720), _F(1.0 / 40320), _F(-1.0 / 3628800)]


def _log_f32(u: np.ndarray) -> np.ndarray:
    """ln(u) for normal positive fp32 u, fp32 ops only."""
    bits = u.view(np.uint32)
    e = ((bits >> np.uint32(23)) & np.uint32(0xFF)).astype(np.int32) - 127
    m = ((bits & np.uint32(0x007FFFFF)) | np.uint32(0x3F800000)).view(np.float32)  # [1,2)
    big = m > _SQRT2
    m = np.where(big, m * _F(0.5), m).astype(np.float32)
    e = np.where(big, e + 1, e)
    f = (m - _F(1.0)).astype(np.float32)
    s = (f / (_F(2.0) + f)).astype(np.float32)
    s2 = (s * s).astype(np.float32)
    poly = _LOG_C[5]
    for k in (4, 3, 2, 1, 0):
        poly = ((poly * s2).astype(np.float32) + _LOG_C[k]).astype(np.float32)
    lm = (poly * s).astype(np.float32)
    return ((e.astype(np.float32) * _LN2).astype(np.float32) + lm).astype(np.float32)


def _sqrt_f32(x: np.ndarray) -> np.ndarray:
    """correctly rounded fp32 square root (IEEE: numpy's sqrt on float32 is; the kernel's sqrtf is too); x <= 0 -> 0."""
    pos = x > _F(0.0)
    xs = np.where(pos, x, _F(1.0)).astype(np.float32)
    return np.where(pos, np.sqrt(xs, dtype=np.float32), _F(0.0)).astype(np.float32)


def _sincos_quarter(r: np.ndarray):
    """sin and cos of (pi/2)*r for r in [0,1), fp32 ops only."""
    swap = r > _F(0.5)
    rr = np.where(swap, (_F(1.0) - r).astype(np.float32), r).astype(np.float32)
    x = (rr * _HALF_PI).astype(np.float32)
    x2 = (x * x).astype(np.float32)
    ps = _SIN_C[3]
    for k in (2, 1, 0):
        ps = ((ps * x2).astype(np.float32) + _SIN_C[k]).astype(np.float32)
    sn = (x + ((x * x2).astype(np.float32) * ps).astype(np.float32)).astype(np.float32)
    pc = _COS_C[4]
    for k in (3, 2, 1, 0):
        pc = ((pc * x2).astype(np.float32) + _COS_C[k]).astype(np.float32)
    cs = (_F(1.0) + (x2 * pc).astype(np.float32)).astype(np.float32)
    return np.where(swap, cs, sn).astype(np.float32), np.where(swap, sn, cs).astype(np.float32)


def normal_pair_from_bits(b0: np.ndarray, b1: np.ndarray):
    """Box-Muller: u1 = ((b0>>8)+1)*2^-24 in (0,1], u2 = (b1>>8)*2^-24 in [0,1);
    (z0, z1) = sqrt(-2 ln u1) * (cos, sin)(2 pi u2)."""
    u1 = ((b0 >> np.uint32(8)).astype(np.float32) + _F(1.0)) * _F(2.0 ** -24)
    u1 = u1.astype(np.float32)
    t = (b1 >> np.uint32(8))
    quad = (t >> np.uint32(22)).astype(np.int32)                  # 2 top bits of 24
    r = ((t & np.uint32(0x3FFFFF)).astype(np.float32) * _F(2.0 ** -22)).astype(np.float32)
    sn, cs = _sincos_quarter(r)
    # rotate by quadrant: angle = (quad + r) * pi/2
    c = np.select([quad == 0, quad == 1, quad == 2], [cs, -sn, -cs], default=sn).astype(np.float32)
    s = np.select([quad == 0, quad == 1, quad == 2], [sn, cs, -sn], default=-cs).astype(np.float32)
    rad = _sqrt_f32((_F(-2.0) * _log_f32(u1)).astype(np.float32))
    return (rad * c).astype(np.float32), (rad * s).astype(np.float32)


def philox_normal(n: int, seed: int, stream: int, step: int, rank: int = 0) -> np.ndarray:
    """eps element i: words (0,1) of block i/4 give elements 4b, 4b+1; words (2,3) give 4b+2, 4b+3."""
    nb = (n + 3) // 4
    bits = philox_bits(nb * 4, seed, stream, step, rank).reshape(nb, 4)
    z0, z1 = normal_pair_from_bits(bits[:, 0], bits[:, 1])
    z2, z3 = normal_pair_from_bits(bits[:, 2], bits[:, 3])
    return np.stack([z0, z1, z2, z3], axis=1).reshape(-1)[:n]


# ----------------------------------------------------------------------------
# Test-input helpers shared by oracle/make_fixtures.py and tests/
# ----------------------------------------------------------------------------
def perturb_norm_params(sd, seed: int):
    """Deterministically move every BatchNorm affine parameter / running stat
    away from its (1, 0, ~0, ~1) initial value so that parity tests exercise
    them.  Uses a private generator; the global RNG is untouched."""
    g = torch.Generator().manual_seed(seed + 1)
    for k in sd:
        if k.endswith("num_batches_tracked") or sd[k].dim() != 1:
            continue
        is_bn = (".net.1." in k or ".net.4." in k or k.startswith("bn"))
        if not is_bn:
            continue
        n = sd[k].numel()
        if k.endswith(".weight"):
            sd[k] = 0.5 + torch.rand(n, generator=g)
        elif k.endswith(".bias"):
            sd[k] = 0.4 * torch.rand(n, generator=g) - 0.2
        elif k.endswith("running_mean"):
            sd[k] = 0.2 * torch.rand(n, generator=g) - 0.1
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(n, generator=g)
    return sd


def make_inputs(n, in_channels, image_size, latent_dims, seed):
    g = torch.Generator().manual_seed(seed + 2)
    x = torch.rand(n, in_channels, image_size, image_size, generator=g)
    eps = torch.randn(n, latent_dims, generator=g)
    return x, eps


def tensor_digest(t: torch.Tensor, nsample: int = 256):
    """(l2 norm, sum, strided samples) of a tensor, for compact golden files."""
    f = t.detach().reshape(-1).double()
    stride = max(1, f.numel() // nsample)
    return (np.array([f.norm().item(), f.sum().item()], dtype=np.float64),
            f[::stride][:nsample].float().numpy().copy())


# ----------------------------------------------------------------------------
# bf16-faithful variant of the forward pass.
# The MI355X path feeds its matrix cores bf16 operands (fp32 accumulate) and keeps
# activations / activation gradients in bf16 between kernels.  This restatement
# rounds at exactly those points, forward and backward, so that the HIP path can
# be checked against it far more tightly (fp32 summation order only) than against
# the fp32 reference, from which any bf16 pipeline of this depth differs by the
# accumulated rounding noise (~3 % at the encoder output for random weights).
# ----------------------------------------------------------------------------
def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


class _RoundFwd(torch.autograd.Function):
    """bf16 rounding in the forward pass, identity in the backward pass."""
    @staticmethod
    def forward(ctx, x):
        return _bf(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    """identity in the forward pass, bf16 rounding of the gradient in the backward pass."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf(g)


def _rf(x):
    return _RoundFwd.apply(x)


def _rb(x):
    return _RoundBwd.apply(x)


def _bn_bf16(y, gamma, beta, rm, rv, train, stats_out, prefix):
    """statistics from the fp32 conv output, normalisation applied to its bf16 copy."""
    if train:
        n = y.numel() // y.shape[1]
        mean = y.mean(dim=(0, 2, 3))
        var = y.var(dim=(0, 2, 3), unbiased=False)
        if stats_out is not None:
            with torch.no_grad():
                stats_out[f"{prefix}.running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean
                stats_out[f"{prefix}.running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * var * (n / max(n - 1, 1))
    else:
        mean, var = rm, rv
    scale = gamma * torch.rsqrt(var + BN_EPS)
    shift = beta - mean * scale
    ys = _rb(_rf(y))          # stored bf16; its gradient is stored bf16 too
    return ys * scale[None, :, None, None] + shift[None, :, None, None]


def forward_bf16(x, sd, eps, train=True, p_dropout=0.0, masks=None, stats_out=None):
    """forward() with the MI355X path's bf16 rounding points (see above)."""
    masks = masks or [None] * 4
    h = _bf(x)
    for i in range(1, 5):
        for j in (0, 3):
            w = _rf(sd[f"conv{i}.net.{j}.weight"])
            y = F.conv2d(h, w, sd[f"conv{i}.net.{j}.bias"], padding=1)
            b = f"conv{i}.net.{j + 1}"
            a = torch.relu(_bn_bf16(y, sd[f"{b}.weight"], sd[f"{b}.bias"], sd[f"{b}.running_mean"],
                                    sd[f"{b}.running_var"], train, stats_out, b))
            if j == 3:
                a = dropout(a, p_dropout, train, masks[i - 1])
                if i <= 3:
                    a = F.max_pool2d(a, 2)
            h = _rb(_rf(a))
    flat = torch.flatten(h, start_dim=1)
    mu = F.linear(flat, _rf(sd["fc_mu.weight"]), sd["fc_mu.bias"])
    logvar = F.linear(flat, _rf(sd["fc_logvar.weight"]), sd["fc_logvar.bias"])
    # the head's input gradient is handed over as bf16 [dmu | dlogvar]
    z = _rf(reparameterize(_rb(mu), _rb(logvar), eps))
    nflat = sd["fc_up.weight"].shape[0]
    hw = int(round(math.sqrt(nflat // 256)))
    pre = _rb(F.linear(z, _rf(sd["fc_up.weight"]), sd["fc_up.bias"]))
    h = _rb(_rf(torch.tanh(pre))).view(-1, 256, hw, hw)
    for k, (up, bn) in enumerate((("up3", "bn3"), ("up2", "bn2"), ("up1", "bn1"))):
        y = F.conv_transpose2d(h, _rf(sd[f"{up}.weight"]), sd[f"{up}.bias"], stride=2)
        a = torch.relu(_bn_bf16(y, sd[f"{bn}.weight"], sd[f"{bn}.bias"], sd[f"{bn}.running_mean"],
                                sd[f"{bn}.running_var"], train, stats_out, bn))
        h = _rb(_rf(a)) if k < 2 else a      # the 1x1 head reads relu(bn1(.)) without storing it
    recon = torch.sigmoid(F.conv2d(h, sd["head.weight"], sd["head.bias"]))
    return recon, mu, logvar


def loss_and_grads_bf16(x, sd, eps, kind="l1", kl_weight=1.0, p_dropout=0.0, masks=None):
    """loss_and_grads() through forward_bf16 (train mode)."""
    params = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if is_param(k)}
    full = dict(sd)
    full.update(params)
    stats = {}
    recon, mu, logvar = forward_bf16(x, full, eps, True, p_dropout, masks, stats)
    loss = vae_loss(recon, x, mu, logvar, kind, kl_weight)
    gl = torch.autograd.grad(loss, list(params.values()), allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(params[k]))
             for k, g in zip(params.keys(), gl)}
    return loss.detach(), recon.detach(), mu.detach(), logvar.detach(), grads, stats


# ----------------------------------------------------------------------------
# The drafts' U-Net VAE (vae/practice/vae_original.py:77-181, class VAEUnet256):
# three GroupNorm double-conv encoder stages whose UNPOOLED outputs are kept as
# skip tensors, a middle block, the latent heads, and three decoder stages
# ConvTranspose2d(k2,s2) -> cat([up, skip], dim=1) -> double conv; the 1x1 head
# returns logits.
# ----------------------------------------------------------------------------
UNET_ENC = (("conv0", None, 64), ("conv1", 64, 128), ("conv2", 128, 256), ("middle", 256, 256))
UNET_DEC = (("up2", 256, 128, "refine2", 128 + 256, 128), ("up1", 128, 64, "refine1", 64 + 128, 64),
            ("up0", 64, 64, "refine0", 64 + 64, 64))
# the five-level sibling, class VAEUnet1024 (vae_original.py:184-403; the second of its two __init__ bodies, :295-340,
# is the one Python keeps): default init of the latent heads, logvar clamped to [-5, 5] before use and return (:390)
UNET1024_ENC = (("conv0", None, 64), ("conv1", 64, 128), ("conv2", 128, 256), ("conv3", 256, 512), ("middle", 512, 1024))
UNET1024_DEC = (("up3", 1024, 512, "refine3", 1024, 512), ("up2", 512, 256, "refine2", 512, 256),
                ("up1", 256, 128, "refine1", 256, 128), ("up0", 128, 64, "refine0", 128, 64))
UNET_VARIANTS = {"256": (UNET_ENC, UNET_DEC, True, None), "1024": (UNET1024_ENC, UNET1024_DEC, False, (-5.0, 5.0))}


def unet_state_shapes(variant="256", latent_dims=64, image_size=32):
    """state_dict names and shapes of VAEUnet256 / VAEUnet1024, in registration order (vae_original.py:88-119,305-340).
    conv0 takes 3 channels whatever ``in_channels`` says (:88)."""
    enc, dec, _, _ = UNET_VARIANTS[variant]
    s = image_size >> (len(enc) - 1)
    flat = enc[-1][2] * s * s
    shapes = OrderedDict()

    def block(name, ci, co):
        shapes[f"{name}.double_conv.0.weight"] = (co, ci, 3, 3)
        shapes[f"{name}.double_conv.0.bias"] = (co,)
        shapes[f"{name}.double_conv.1.weight"] = (co,)
        shapes[f"{name}.double_conv.1.bias"] = (co,)
        shapes[f"{name}.double_conv.3.weight"] = (co, co, 3, 3)
        shapes[f"{name}.double_conv.3.bias"] = (co,)
        shapes[f"{name}.double_conv.4.weight"] = (co,)
        shapes[f"{name}.double_conv.4.bias"] = (co,)

    for name, ci, co in enc:
        block(name, ci or 3, co)
    for name in ("fc_mu", "fc_logvar"):
        shapes[f"{name}.weight"] = (latent_dims, flat)
        shapes[f"{name}.bias"] = (latent_dims,)
    shapes["fc_up.weight"] = (flat, latent_dims)
    shapes["fc_up.bias"] = (flat,)
    for up, ci, co, ref, rci, rco in dec:
        shapes[f"{up}.weight"] = (ci, co, 2, 2)
        shapes[f"{up}.bias"] = (co,)
        block(ref, rci, rco)
    shapes["head.weight"] = (3, 64, 1, 1)
    shapes["head.bias"] = (3,)
    return shapes


def construct_state_unet(variant="256", latent_dims=64, p_dropout=0.1, image_size=32):
    """Restates the constructors (vae_original.py:78-119, 295-340), consuming torch's global CPU generator in the same
    order: the encoder blocks, the train-mode dummy forward of _get_flattened_shape (:122-138; one Bernoulli mask per
    block when p > 0, GroupNorm keeps no running state), fc_mu, fc_logvar, for VAEUnet256 their
    xavier_uniform_(gain=0.1) (:104-105), fc_up, then up / refine pairs and the head."""
    enc, dec, xavier, _ = UNET_VARIANTS[variant]
    shapes = unet_state_shapes(variant, latent_dims, image_size)
    sd = OrderedDict()

    def conv_like(wname, bname):
        shp = shapes[wname]
        fan_in = shp[1] * (shp[2] * shp[3] if len(shp) == 4 else 1)
        bound = 1.0 / math.sqrt(fan_in)
        sd[wname] = _uniform(shp, bound)
        sd[bname] = _uniform(shapes[bname], bound)

    def block(name):
        for j in (0, 3):
            conv_like(f"{name}.double_conv.{j}.weight", f"{name}.double_conv.{j}.bias")
            c = shapes[f"{name}.double_conv.{j + 1}.weight"][0]
            sd[f"{name}.double_conv.{j + 1}.weight"] = torch.ones(c)
            sd[f"{name}.double_conv.{j + 1}.bias"] = torch.zeros(c)

    for name, _, _ in enc:
        block(name)
    if p_dropout > 0:
        s = image_size
        for i, (name, _, co) in enumerate(enc):
            F.dropout(torch.zeros(1, co, s, s), p_dropout, True)      # the dummy forward's mask draw
            if i < len(enc) - 1:
                s //= 2
    for name in ("fc_mu", "fc_logvar"):
        conv_like(f"{name}.weight", f"{name}.bias")
    if xavier:
        for name in ("fc_mu", "fc_logvar"):
            shp = shapes[f"{name}.weight"]
            sd[f"{name}.weight"] = _uniform(shp, 0.1 * math.sqrt(6.0 / (shp[0] + shp[1])))
    conv_like("fc_up.weight", "fc_up.bias")
    for up, _, _, ref, _, _ in dec:
        conv_like(f"{up}.weight", f"{up}.bias")
        block(ref)
    conv_like("head.weight", "head.bias")
    return OrderedDict((k, sd[k]) for k in shapes)


def unet_forward(variant, x, sd, eps, train=True, p_dropout=0.0, masks=None):
    """VAEUnet256.forward (vae_original.py:141-175) / VAEUnet1024.forward (:370-403) with eps injected
    -> (logits, mu, logvar).  masks: optional dict block name -> dropout keep mask."""
    enc, dec, _, clamp = UNET_VARIANTS[variant]
    masks = masks or {}
    skips = []
    for i, (name, _, _) in enumerate(enc):
        x = double_conv_gn(x, sd, name, train, p_dropout, masks.get(name))
        if i < len(enc) - 1:
            skips.append(x)
            x = F.max_pool2d(x, 2)
    n, c, h, w = x.shape
    m = x.reshape(n, c * h * w)
    mu = F.linear(m, sd["fc_mu.weight"], sd["fc_mu.bias"])
    logvar = F.linear(m, sd["fc_logvar.weight"], sd["fc_logvar.bias"])
    if clamp is not None:
        logvar = torch.clamp(logvar, min=clamp[0], max=clamp[1])
    z = reparameterize(mu, logvar, eps)
    x = torch.tanh(F.linear(z, sd["fc_up.weight"], sd["fc_up.bias"])).view(n, c, h, w)
    for up, _, _, ref, _, _ in dec:
        u = F.conv_transpose2d(x, sd[f"{up}.weight"], sd[f"{up}.bias"], stride=2)
        x = double_conv_gn(torch.cat([u, skips.pop()], dim=1), sd, ref, train, p_dropout, masks.get(ref))
    return F.conv2d(x, sd["head.weight"], sd["head.bias"]), mu, logvar


def unet256_state_shapes(latent_dims=64, image_size=32):
    return unet_state_shapes("256", latent_dims, image_size)


def construct_state_unet256(latent_dims=64, p_dropout=0.1, image_size=32):
    return construct_state_unet("256", latent_dims, p_dropout, image_size)


def unet256_forward(x, sd, eps, train=True, p_dropout=0.0, masks=None):
    return unet_forward("256", x, sd, eps, train, p_dropout, masks)


def unet_loss_mse(x, x_hat, mu, logvar, kl_weight):
    """vae_loss_mse of the draft script (vae_original.py:527-532): squared error on the head's logits."""
    return (((x_hat - x) ** 2).sum() + kl_weight * kl_sum(mu, logvar)) / x.size(0)


def perturb_gn_params(sd, seed: int):
    """moves every GroupNorm affine away from its (1, 0) init, deterministically in key order (so that fixtures
    need not store the weights: construct_state_unet256 under the seed + this reproduce them)."""
    g = torch.Generator().manual_seed(seed + 1)
    for k in sd:
        if ".double_conv.1." in k or ".double_conv.4." in k:
            sd[k] = sd[k] + 0.3 * torch.randn(sd[k].shape, generator=g)
    return sd
