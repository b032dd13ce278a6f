"""Clip + AdamW on the model's flat buffers (two kernel launches per step).

Same update rule and defaults as the reference's optimizer
(torch.optim.AdamW(model.parameters(), lr), vae/training/train.py:310: betas (0.9, 0.999),
eps 1e-8, weight_decay 0.01) preceded by clip_grad_norm_(max_norm) (train.py:336)."""
from __future__ import annotations

import torch

from . import _lib


class FlatAdamW:
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        if model.flat_params is None:
            raise RuntimeError("move the model to the GPU before creating the optimizer")
        self.model = model
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), betas, float(eps), float(weight_decay)
        self.exp_avg = torch.zeros_like(model.flat_params)
        self.exp_avg_sq = torch.zeros_like(model.flat_params)
        self.step_count = 0
        n = model.flat_params.numel()
        self._partial = torch.empty(_lib.call("vae_adamw_grid", n), device=model.flat_params.device)
        self.grad_norm = torch.zeros(1, device=model.flat_params.device)   # pre-clip norm of the last step
        self.grad_mul = 1.0          # gradients are multiplied by this first (1/world_size after a SUM all-reduce)

    def zero_grad(self, set_to_none=True):
        for p in self.model.parameters():
            p.grad = None

    def step_flat(self, model=None, max_norm=1.0, elbo=None):
        """clip_grad_norm_(max_norm) + AdamW step on model.flat_grads, then refresh the bf16 weights.
        elbo: the arguments of vae_elbo_combine (recon partials, count, KL partials, count, kl_weight, 1/batch, out3);
        the step's loss sums are then finished inside the first launch here instead of in a launch of their own."""
        m = self.model
        _lib.check_device(m.flat_params)
        self.step_count += 1
        e = elbo if elbo is not None else (None, 0, None, 0, 0.0, 0.0, None)
        _lib.call("vae_adamw_clip_step_elbo", m.flat_params.data_ptr(), m.flat_grads.data_ptr(), self.exp_avg.data_ptr(),
                  self.exp_avg_sq.data_ptr(), m.flat_params.numel(), self._partial.data_ptr(), float(self.grad_mul),
                  float(max_norm if max_norm is not None else 0.0), self.lr, self.betas[0], self.betas[1], self.eps,
                  self.weight_decay, self.step_count, self.grad_norm.data_ptr(), e[0], int(e[1]), e[2], int(e[3]),
                  float(e[4]), float(e[5]), e[6], _lib.current_stream())
        m.repack_weights()

    def step(self):
        """optimizer.step() on parameters' .grad (no clipping): gathers .grad into flat_grads first."""
        m = self.model
        for p, view in zip(m._ordered_params(), m.grad_views()):
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
        self.step_flat(m, max_norm=None)
