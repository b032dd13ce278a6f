"""MI355X drop-in for the loss and train-step entry points of the reference's training
script (vae/training/train.py:44-55 loss, :324-340 train loop body, :346-355 val body,
:443-453 config loading).  Same function names and argument order; the arithmetic runs in
libvae_hip.so (fused reconstruction-loss forward+backward, KL fused with the
reparameterisation kernels).  The LPIPS perceptual term of the reference (train.py:49) needs
a pretrained VGG and is not part of this path: percep_weight must be 0 / lpips_fn None.
"""
from __future__ import annotations

import os
import re
from typing import Callable, Optional

import torch
import torch.nn.functional as F
import yaml

from . import _lib

# "bce_logits" is the draft script's vae_loss_bce (vae/practice/vae_conv.py:146-151): logits clamped to [-10, 10]
# first, no gradient outside.  The torch callable itself is unclamped ("bce_logits_raw").
RECON_KINDS = {"l1": 0, "mse": 1, "bce": 2, "bce_logits": 3, "bce_logits_raw": 4}
_CALLABLE_KINDS = {F.l1_loss: 0, F.mse_loss: 1, F.binary_cross_entropy: 2, F.binary_cross_entropy_with_logits: 4}


# ---------------------------------------------------------------------- config (train.py:443-453)
class Config(dict):
    """dict with attribute access (the subset of OmegaConf's DictConfig the reference uses:
    config.key reads/writes and ${key} interpolation inside strings)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        if isinstance(v, str) and "${" in v:
            v = re.sub(r"\$\{([^}]+)\}", lambda m: str(self[m.group(1)]), v)
        return v

    def __setattr__(self, k, v):
        self[k] = v


def _fix_scalars(d):
    # PyYAML reads "1e-4" (no dot) as a string; OmegaConf reads it as a float
    for k, v in list(d.items()):
        if isinstance(v, str):
            try:
                d[k] = float(v) if re.fullmatch(r"[+-]?\d+(\.\d*)?[eE][+-]?\d+", v) else v
            except ValueError:
                pass
    return d


def load_config(env="local", config_dir="config"):
    """base.yaml merged with {env}.yaml, env overriding base (train.py:443-453)."""
    with open(os.path.join(config_dir, "base.yaml")) as f:
        cfg = Config(_fix_scalars(yaml.safe_load(f) or {}))
    env_path = os.path.join(config_dir, f"{env}.yaml")
    if os.path.exists(env_path):
        with open(env_path) as f:
            cfg.update(_fix_scalars(yaml.safe_load(f) or {}))
    return cfg


# ---------------------------------------------------------------------- loss (train.py:44-55)
def _kind_of(loss) -> int:
    if isinstance(loss, str):
        return RECON_KINDS[loss]
    if loss in _CALLABLE_KINDS:
        return _CALLABLE_KINDS[loss]
    raise ValueError("reconstruction loss must be one of F.l1_loss, F.mse_loss, F.binary_cross_entropy, "
                     "F.binary_cross_entropy_with_logits or 'l1' | 'mse' | 'bce' | 'bce_logits' | 'bce_logits_raw'")


class _ElboLoss(torch.autograd.Function):
    """(recon_sum + kl_weight * KL) / N with the gradient of the reconstruction term produced
    in the same pass that computes it."""

    @staticmethod
    def forward(ctx, recon, target, mu, logvar, kind, kl_weight, denom=None):
        recon, target = recon.contiguous(), target.contiguous()
        mu, logvar = mu.contiguous(), logvar.contiguous()
        n, batch = recon.numel(), float(denom or recon.size(0))      # denom: what the sum is divided by (default: the batch)
        s = _lib.current_stream()
        grid = _lib.call("vae_recon_loss_grid", n)
        partial = torch.empty(grid, device=recon.device)
        need_grad = ctx.needs_input_grad[0]
        grad = torch.empty_like(recon) if need_grad else None
        _lib.call("vae_recon_loss_fwd_bwd", recon.data_ptr(), target.data_ptr(), _lib.ptr(grad), partial.data_ptr(), n,
                  kind, 1.0 / batch, s)
        nl = mu.numel()
        kgrid = _lib.call("vae_reparam_grid", nl)
        klp = torch.empty(kgrid, device=recon.device)
        # eps_in = mu: any valid buffer; with no z output requested only the KL partials are used
        _lib.call("vae_reparam_kl_fwd", mu.data_ptr(), logvar.data_ptr(), mu.data_ptr(), None, None, None,
                  klp.data_ptr(), nl, 0, 0, 0, s)
        out3 = torch.empty(3, device=recon.device)
        _lib.call("vae_elbo_combine", partial.data_ptr(), grid, klp.data_ptr(), kgrid, float(kl_weight), 1.0 / batch,
                  out3.data_ptr(), s)
        ctx.save_for_backward(grad, mu, logvar)
        ctx.kl_scale = float(kl_weight) / batch
        ctx.parts = out3
        return out3[0].clone()

    @staticmethod
    def backward(ctx, g):
        grad, mu, logvar = ctx.saved_tensors
        d_recon = grad * g if grad is not None else None
        nl = mu.numel()
        dmu, dlv = torch.empty_like(mu), torch.empty_like(logvar)
        _lib.call("vae_reparam_kl_bwd", None, mu.data_ptr(), logvar.data_ptr(), mu.data_ptr(), None, None, ctx.kl_scale,
                  dmu.data_ptr(), dlv.data_ptr(), None, nl, mu.size(-1), 0, 0, 0, _lib.current_stream())
        return d_recon, None, dmu * g, dlv * g, None, None, None


def _require_no_percep(config):
    """The reference adds percep_weight * LPIPS (train.py:49-50).  That term is not computed here, so a config
    that asks for it would silently train on a different loss: refuse it on every loss entry point."""
    pw = config.get("percep_weight", 0.0) if isinstance(config, dict) else getattr(config, "percep_weight", 0.0)
    if float(pw or 0.0) != 0.0:
        raise NotImplementedError(f"percep_weight={pw}: the LPIPS perceptual term (pretrained VGG) is outside the "
                                  "MI355X hot path and is not computed: set percep_weight: 0")


def vae_loss(input, target, mu: torch.Tensor, logvar: torch.Tensor, loss: Callable, lpips_fn, config):
    """reference signature (train.py:44).  `loss` selects the reconstruction term; `lpips_fn` is ignored
    (percep_weight must be 0)."""
    _require_no_percep(config)
    _lib.check_device(input)
    return _ElboLoss.apply(input, target, mu, logvar, _kind_of(loss), float(config.kl_weight))


def l1_kl_percep_loss(recons, inputs, mu, logvar, config, LPIPS_FN=None):
    """reference signature (train.py:54-55)."""
    return vae_loss(recons, inputs, mu, logvar, F.l1_loss, LPIPS_FN, config)


def linear_anneal_kl(config, epoch) -> float:
    """KL weight schedule of the draft training script (vae/practice/vae_conv.py:154-157):
    ``kl_start + (kl_end - kl_start) * min(epoch / n_epochs, kl_clip)``.  The live loop uses the constant
    ``config.kl_weight`` (train.py:50); to anneal, assign the result to ``config.kl_weight`` at the top of each
    epoch -- the loss kernels take the weight as a per-call scalar."""
    progress = min(epoch / config.n_epochs, config.kl_clip)
    return config.kl_start + (config.kl_end - config.kl_start) * progress


def recon_kind_from_config(config) -> int:
    return RECON_KINDS[getattr(config, "recon_loss", "l1") if "recon_loss" in config else "l1"]


# ---------------------------------------------------------------------- train / val step
def train_step(model, optimizer, inputs, config, eps=None, keep_masks=None, max_norm=1.0):
    """The reference's loop body, in its order (train.py:327-339): forward, zero_grad, loss,
    backward, clip_grad_norm_(1.0), optimizer.step().  Goes through torch autograd (the model
    and the loss are one autograd node each).  Returns the loss as a 0-dim device tensor (the
    caller decides when to .item() it)."""
    _require_no_percep(config)
    recons, mu, logvar = model(inputs, eps, keep_masks)
    optimizer.zero_grad()
    kind = recon_kind_from_config(config)
    loss = _ElboLoss.apply(recons, inputs, mu, logvar, kind, float(config.kl_weight))
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=max_norm)
    optimizer.step()
    return loss.detach()


def val_step(model, inputs, config, eps=None):
    """The reference's validation body (train.py:350-353): eval-mode forward + the same loss.
    Like the reference, eps is still sampled in eval mode (vae.py:108-111)."""
    _require_no_percep(config)
    with torch.no_grad():
        recons, mu, logvar = model(inputs, eps)
        kind = recon_kind_from_config(config)
        return _ElboLoss.apply(recons, inputs, mu, logvar, kind, float(config.kl_weight))


class FusedStep:
    """The same train step without the autograd engine: forward, fused loss+dloss, backward in
    stages (so a data-parallel reducer can overlap the all-reduce of finished stages with the
    rest of the backward), then clip + AdamW.  Numerically identical to train_step()."""

    def __init__(self, model, optimizer, config, reducer=None, max_norm=1.0):
        _require_no_percep(config)
        self.model, self.optimizer, self.config, self.reducer, self.max_norm = model, optimizer, config, reducer, max_norm
        self.kind = recon_kind_from_config(config)
        self.kl_weight = float(config.kl_weight)
        self._grad_views = None
        self._bufs = {}

    def _buffers(self, recon, dev):
        key = tuple(recon.shape)
        if key not in self._bufs:
            n = recon.numel()
            grid = _lib.call("vae_recon_loss_grid", n)
            self._bufs[key] = (torch.empty_like(recon), torch.empty(grid, device=dev), torch.empty(3, device=dev), grid)
        return self._bufs[key]

    def __call__(self, inputs, eps=None, keep_masks=None):
        m = self.model
        recon, mu, logvar, state = m.forward_raw(inputs, eps, keep_masks)
        batch = inputs.size(0)
        d_recon, partial, out3, grid = self._buffers(recon, inputs.device)
        s = _lib.current_stream()
        _lib.call("vae_recon_loss_fwd_bwd", recon.data_ptr(), inputs.data_ptr(), d_recon.data_ptr(), partial.data_ptr(),
                  recon.numel(), self.kind, 1.0 / batch, s)
        klp, nkl = m.kl_partials(state)
        elbo = (partial.data_ptr(), grid, klp, nkl, self.kl_weight, 1.0 / batch, out3.data_ptr())
        # the loss value has no reader on the device: FlatAdamW finishes its sums inside its own first launch
        defer = hasattr(self.optimizer, "step_flat")
        if not defer:
            _lib.call("vae_elbo_combine", *elbo, s)
        kl_scale = self.kl_weight / batch
        if self.reducer is None:
            m.backward_raw(state, d_recon, None, None, kl_scale, -1)
        else:
            for stage in (0, 1, 2):
                m.backward_raw(state, d_recon, None, None, kl_scale, stage)
                self.reducer.stage_done(stage)
            self.reducer.wait()
        if self._grad_views is None or self._grad_views[0].data_ptr() != m.flat_grads.data_ptr():
            self._grad_views = m.grad_views()
        if hasattr(self.optimizer, "step_flat"):
            self.optimizer.step_flat(m, self.max_norm, elbo)
        else:
            for p, g in zip(m._ordered_params(), self._grad_views):
                p.grad = g
            torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=self.max_norm)
            self.optimizer.step()
            m.repack_weights()        # the torch optimizer wrote the fp32 parameters: refresh the bf16 operand copies
        # only now: the optimizer's first launch reads the KL partial sums out of this workspace (`elbo`), so the buffer
        # must not go back to the pool (where the next forward of any caller would take it) before that launch is queued
        m._release_workspace(state["batch"], state["ws"])
        return out3[0]


# ---------------------------------------------------------------------- epoch loop
def fit(model, optimizer, config, train_feed, val_feed, log=None, checkpoint_path="best-model.pth", reducer=None,
        fused=True):
    """The reference's epoch loop (train.py:316-373) around the train / val steps above: per epoch a training
    pass (``model.train()``), a validation pass (``model.eval()``, no gradients, eps still sampled), the epoch
    losses as the mean of the per-batch losses, and -- when ``config.save_model`` -- ``best-model.pth`` rewritten
    whenever the validation loss improves (train.py:360-367).  ``log`` receives what the reference hands to
    ``wandb.log`` (train.py:369-373); the list of those records is returned.

    ``train_feed`` / ``val_feed`` are iterables of device batches (``vae_amd.data.DeviceFeed``, or a DataLoader
    whose batches are moved with ``.to(device)``; ``(inputs, label)`` pairs are accepted).  The per-batch losses
    stay on the device and are read once per epoch (the reference's ``loss.item()`` every step stalls the stream;
    the numbers are the same).  ``fused=True`` runs ``FusedStep`` (requires ``FlatAdamW``); otherwise the
    autograd ``train_step``."""
    import torch.distributed as dist
    from .data import save_checkpoint
    from .dp import sync_bn_stats
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    step = FusedStep(model, optimizer, config, reducer) if fused else None
    best_val_loss = float("inf")
    history = []

    def batches(feed):
        for b in feed:
            yield b[0] if isinstance(b, (tuple, list)) else b

    for epoch in range(1, int(config.n_epochs) + 1):
        model.train()
        losses = [(step(x) if fused else train_step(model, optimizer, x, config)).clone() for x in batches(train_feed)]
        train_epoch_loss = float(torch.stack(losses).sum()) / max(len(losses), 1)
        if multi:
            # BatchNorm batch statistics are per rank, so the running statistics drift apart during the epoch: average
            # them over the ranks first, so that every rank validates -- and rank 0 saves -- the SAME model
            sync_bn_stats(model)
        model.eval()
        with torch.no_grad():
            losses = [val_step(model, x, config).clone() for x in batches(val_feed)]
        val_epoch_loss = float(torch.stack(losses).sum()) / max(len(losses), 1) if losses else float("nan")
        if multi:
            # Data-parallel ranks see different validation shards.  The epoch losses become the mean over ranks, so
            # every rank takes the same "improved" decision, and ONE rank writes the file: parameters are replicated
            # and the running statistics were averaged before the validation pass, so the file is the model whose
            # loss was compared.
            t = torch.tensor([train_epoch_loss, val_epoch_loss], dtype=torch.float64,
                             device=next(model.parameters()).device)
            dist.all_reduce(t)
            train_epoch_loss, val_epoch_loss = (t / dist.get_world_size()).tolist()
        if getattr(config, "save_model", False) and val_epoch_loss < best_val_loss:
            best_val_loss = val_epoch_loss
            if not multi or dist.get_rank() == 0:
                save_checkpoint(model, checkpoint_path)
        rec = {"epoch": epoch, "train/loss": train_epoch_loss, "val/loss": val_epoch_loss}
        history.append(rec)
        if log is not None:
            log(rec)
    return history
