"""MI355X drop-in for the reference's conv VAE module (vae/models/vae.py:22-141).

Same class names, constructor signature (positional order included), attribute names
(conv1..conv4, pool1..pool3, fc_mu, fc_logvar, fc_up, up3/bn3, up2/bn2, up1/bn1, head, chw,
flattened_dims, image_size), methods (encode / reparameterize / decode / forward) and the
same 85-entry state_dict, so reference checkpoints load and reference call sites run
unchanged.  What differs is underneath: forward and backward are each ONE call into
libvae_hip.so (hand-written gfx950 kernels); there is no torch / CPU compute fallback -- a
CPU tensor or a missing library raises.

Storage: once the module sits on a GPU all 52 parameters are views into one flat fp32
buffer (`flat_params`), gradients are produced into a second flat buffer of the same layout
(`flat_grads`) and BatchNorm running stats live in a third; parameter identity, shapes and
names are unchanged.  The kernels read bf16 packed copies of the weights; they are refreshed
before a pass whenever any parameter's (or the flat buffer's) autograd version counter has
moved since the last pack -- torch optimizers, load_state_dict, p.copy_() and p.add_() all
bump the counter of the Parameter they write, not the flat buffer's.

Precision: ``model.set_precision("fp32")`` switches the same module (same buffers, same entry
points) to the library's fp32 kernels -- the reference's own arithmetic type
(vae/training/train.py:430-431): NCHW fp32 activations, parameters read unpacked, sums in fp64.
That mode reproduces the reference to fp32 rounding, gradients included; the default "bf16"
mode is the throughput path (bf16 matrix-core operands, fp32 accumulation).
"""
from __future__ import annotations

import ctypes
import math
from typing import Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib

__all__ = ["VAEConv", "DoubleConv"]


class DoubleConv(nn.Module):
    """Parameter container with the reference's layout (vae/models/vae.py:125-141):
    net = [Conv2d 3x3 p1, BatchNorm2d, ReLU, Conv2d 3x3 p1, BatchNorm2d, ReLU, Dropout].
    It is executed by VAEConv (fused HIP kernels), never on its own."""

    def __init__(self, in_channels, out_channels, p_dropout=0.1):
        super().__init__()
        self.net = nn.Sequential(
            nn.Conv2d(in_channels, out_channels, 3, padding=1),
            nn.BatchNorm2d(num_features=out_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, 3, padding=1),
            nn.BatchNorm2d(num_features=out_channels),
            nn.ReLU(inplace=True),
            nn.Dropout(p=p_dropout),
        )

    def forward(self, x):
        raise RuntimeError("DoubleConv is executed as part of VAEConv.forward/encode on the GPU; "
                           "it has no standalone (torch) forward")


def _construction_dummy_forward(blocks: Sequence[DoubleConv], in_channels: int, image_size: int):
    """The reference constructor pushes zeros(1,C,S,S) through the encoder in TRAIN mode to
    measure the flattened size (vae/models/vae.py:78-86).  That moves every encoder
    BatchNorm's running stats one step (num_batches_tracked = 1) and draws one dropout mask
    per block from torch's global generator.  Reproduced here once, at construction, with the
    same ATen calls so that a seeded construction equals the reference's bit for bit."""
    with torch.no_grad():
        x = torch.zeros(1, in_channels, image_size, image_size)
        for i, blk in enumerate(blocks):
            x = blk.net(x)
            if i < 3:
                x = F.max_pool2d(x, 2)
        return tuple(x.shape[1:])


class VAEConv(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, latent_dims=128, p_dropout=0.1, image_size=224):
        super().__init__()
        self.image_size = image_size
        self.conv1 = DoubleConv(in_channels, 64, p_dropout)
        self.pool1 = nn.MaxPool2d(2)
        self.conv2 = DoubleConv(64, 128, p_dropout)
        self.pool2 = nn.MaxPool2d(2)
        self.conv3 = DoubleConv(128, 256, p_dropout)
        self.pool3 = nn.MaxPool2d(2)
        self.conv4 = DoubleConv(256, 256, p_dropout)
        C, H, W = _construction_dummy_forward([self.conv1, self.conv2, self.conv3, self.conv4],
                                              in_channels, image_size)
        self.flattened_dims = C * H * W
        self.chw = C, H, W
        self.fc_mu = nn.Linear(self.flattened_dims, latent_dims)
        self.fc_logvar = nn.Linear(self.flattened_dims, latent_dims)
        nn.init.xavier_uniform_(self.fc_mu.weight, gain=0.1)        # vae.py:60-61
        nn.init.xavier_uniform_(self.fc_logvar.weight, gain=0.1)
        self.fc_up = nn.Linear(latent_dims, self.flattened_dims)
        self.up3 = nn.ConvTranspose2d(256, 128, 2, stride=2)
        self.bn3 = nn.BatchNorm2d(num_features=128)
        self.up2 = nn.ConvTranspose2d(128, 64, 2, stride=2)
        self.bn2 = nn.BatchNorm2d(num_features=64)
        self.up1 = nn.ConvTranspose2d(64, 64, 2, stride=2)
        self.bn1 = nn.BatchNorm2d(num_features=64)
        self.head = nn.Conv2d(64, out_channels, 1, stride=1)

        # ---- MI355X runtime state (plain attributes: not part of the state_dict)
        self._hp = dict(in_channels=in_channels, out_channels=out_channels, latent_dims=latent_dims,
                        p_dropout=float(p_dropout), image_size=image_size)
        self.rng_seed = 1337        # Philox key; config.seed in the reference's yaml (config/base.yaml:1)
        self.rng_step = 0           # Philox counter word 1: index of the forward call
        self.rank = 0               # Philox counter word 3: data-parallel rank
        self.flat_params = None
        self.flat_grads = None
        self._bn_stats = None
        self._nbt = None
        self._wpack = None
        self._packed_version = None
        self._ws_pool = {}
        self._last = None           # state of the most recent forward (for the fused train step)
        # True: backward runs the weight-gradient kernels on a second stream beside the dgrad / BatchNorm chain
        # (measured +0.5 % at batch 512, 32x32: the chain leaves few idle CUs; off by default)
        self.use_side_stream = False
        self._side_stream = None
        self.precision = "bf16"     # or "fp32": see set_precision

    def set_precision(self, precision: str):
        """"bf16" (default): bf16 MFMA kernels.  "fp32": the library's fp32 kernels, which match the fp32 reference
        to rounding (outputs, loss and every gradient).  Parameters, optimizer state and checkpoints are fp32 in
        both modes, so the mode can change between any two passes."""
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {precision!r}")
        self.precision = precision
        return self

    # ------------------------------------------------------------------ layout helpers
    def _ordered_params(self):
        """the 52 parameters in state_dict order (== the order of the C library's layout)."""
        return [p for _, p in self.named_parameters()]

    def _bn_modules(self):
        out = []
        for blk in (self.conv1, self.conv2, self.conv3, self.conv4):
            out += [blk.net[1], blk.net[4]]
        return out + [self.bn3, self.bn2, self.bn1]

    def _cfg(self, batch: int, train: bool, step: Optional[int] = None) -> "_lib.ModelCfg":
        hp = self._hp
        side = None
        if train and self.use_side_stream and self.flat_params is not None and self.flat_params.is_cuda:
            if self._side_stream is None or self._side_stream.device != self.flat_params.device:
                self._side_stream = torch.cuda.Stream(device=self.flat_params.device)
            side = self._side_stream.cuda_stream
        return _lib.ModelCfg(batch=batch, in_channels=hp["in_channels"], out_channels=hp["out_channels"],
                             latent=hp["latent_dims"], image_size=hp["image_size"], train=int(train),
                             p_dropout=hp["p_dropout"], seed=int(self.rng_seed),
                             step=int(self.rng_step if step is None else step) & 0xFFFFFFFF, rank=int(self.rank),
                             side_stream=side, precision=_lib.PRECISIONS[self.precision],
                             batches_tracked=self._nbt_ptr(train))

    def _nbt_ptr(self, train: bool):
        """device address of the eleven num_batches_tracked counters when the library bumps them itself (train mode,
        bf16 path, counters in one int64 tensor on the device), else None: then the Python side adds 1"""
        t = self._nbt
        if train and self.precision == "bf16" and t is not None and t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() \
                and t.numel() == 11:
            return t.data_ptr()
        return None

    def _flatten(self, device):
        """Move parameters / running stats into the flat buffers the kernels use (views keep
        every Parameter object, name and shape as they were)."""
        lib = _lib.load()
        cfg = self._cfg(1, True)
        total = lib.vae_model_param_count(ctypes.byref(cfg))
        if total < 0:
            raise RuntimeError(f"unsupported VAEConv configuration: {lib.vae_last_error().decode()}")
        params = self._ordered_params()
        assert len(params) == 52, len(params)
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        grads = torch.zeros(total, dtype=torch.float32, device=device)
        self._offsets = []
        with torch.no_grad():
            for i, p in enumerate(params):
                off = lib.vae_model_param_offset(ctypes.byref(cfg), i)
                n = p.numel()
                self._offsets.append((off, n))
                view = flat[off:off + n].view(p.shape)
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
                p.data = view
                p.grad = None
        bns = self._bn_modules()
        nbn = lib.vae_model_bn_count(ctypes.byref(cfg))
        stats = torch.zeros(nbn, dtype=torch.float32, device=device)
        nbt = torch.zeros(len(bns), dtype=torch.int64, device=device)
        o = 0
        with torch.no_grad():
            for i, bn in enumerate(bns):
                C = bn.num_features
                rm, rv = stats[o:o + C], stats[o + C:o + 2 * C]
                rm.copy_(bn.running_mean.to(device))
                rv.copy_(bn.running_var.to(device))
                nbt[i] = bn.num_batches_tracked.to(device)
                bn.running_mean, bn.running_var, bn.num_batches_tracked = rm, rv, nbt[i]
                o += 2 * C
        self.flat_params, self.flat_grads, self._bn_stats, self._nbt = flat, grads, stats, nbt
        # zero-filled once: the padding of the packed first-conv operand is never rewritten
        self._wpack = torch.zeros(lib.vae_model_wpack_bytes(ctypes.byref(cfg)), dtype=torch.uint8, device=device)
        self._packed_version = None
        self._param_list = params
        self._ws_pool = {}
        self._last = None

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        dev = self.head.weight.device
        if dev.type == "cuda":
            self._flatten(dev)
        else:
            self.flat_params = self.flat_grads = self._bn_stats = self._nbt = self._wpack = None
            self._ws_pool = {}
        return self

    def grad_views(self, base: Optional[torch.Tensor] = None):
        """per-parameter views (reference shapes) of `base` (default: flat_grads)."""
        base = self.flat_grads if base is None else base
        return [base[off:off + n].view(p.shape) for (off, n), p in zip(self._offsets, self._ordered_params())]

    def _weights_signature(self):
        """Changes whenever torch code has written a parameter since the last pack.  Every in-place write through
        a Parameter (optimizer.step(), load_state_dict, p.copy_(), p.add_()) bumps THAT tensor's version counter
        -- the views were attached with `p.data = view`, so they do not share the flat buffer's counter -- and a
        write through flat_params bumps the flat buffer's.  A re-assigned `.data` shows up as a moved pointer."""
        ps = self._param_list
        return (self.flat_params._version, sum(p._version for p in ps), ps[0].data_ptr(), ps[-1].data_ptr())

    def repack_weights(self):
        """refresh the bf16 operand copies from flat_params (call after a raw-pointer update).  The fp32 kernels read
        the parameters themselves: in that mode the copies are only marked stale."""
        if self.precision != "bf16":
            self._packed_version = None
            return
        cfg = self._cfg(1, True)
        _lib.check_device(self.flat_params)
        _lib.call("vae_model_pack_weights", ctypes.byref(cfg), self.flat_params.data_ptr(), self._wpack.data_ptr(),
                  _lib.current_stream())
        self._packed_version = self._weights_signature()

    def _ensure_ready(self, x):
        if not x.is_cuda:
            raise RuntimeError("VAEConv runs on an MI355X only: move the module and its input to 'cuda' "
                               "(there is no CPU execution path)")
        _lib.check_device(x)
        if self.flat_params is None or self.flat_params.device != x.device:
            raise RuntimeError("module parameters are not on the input's device: call model.to(inputs.device)")
        if x.dtype != torch.float32:
            raise RuntimeError(f"inputs must be float32 in [0,1] like the reference's ToTensor() output, got {x.dtype}")
        if self._param_list[0].data_ptr() != self.flat_params.data_ptr():
            raise RuntimeError("a parameter no longer lives in the module's flat buffer (its .data was re-assigned, e.g. "
                               "load_state_dict(assign=True)): call model.to(device) again to re-attach the parameters")
        if self.precision == "bf16" and self._packed_version != self._weights_signature():
            self.repack_weights()

    def _workspace(self, batch):
        cfg = self._cfg(batch, True)
        nbytes = _lib.load().vae_model_workspace_bytes(ctypes.byref(cfg))     # depends on batch and precision
        if nbytes < 0:
            raise RuntimeError(f"unsupported VAEConv configuration: {_lib.load().vae_last_error().decode()}")
        pool = self._ws_pool.setdefault(nbytes, [])
        if pool:
            return pool.pop()
        return torch.empty(nbytes, dtype=torch.uint8, device=self.flat_params.device)

    def _release_workspace(self, batch, ws):
        pool = self._ws_pool.setdefault(ws.numel(), [])
        if len(pool) < 2:
            pool.append(ws)

    @staticmethod
    def _mask_ptrs(keep_masks):
        if keep_masks is None:
            return None, None
        if len(keep_masks) != 4:
            raise ValueError("keep_masks must hold 4 uint8 NHWC masks, one per DoubleConv")
        ms = [m.contiguous() for m in keep_masks]
        for m in ms:
            if m.dtype != torch.uint8 or not m.is_cuda:
                raise ValueError("keep_masks must be uint8 CUDA tensors in NHWC layout")
        arr = (ctypes.c_void_p * 4)(*[m.data_ptr() for m in ms])
        return arr, ms

    # ------------------------------------------------------------------ raw (non-autograd) passes
    def forward_raw(self, x, eps=None, keep_masks=None):
        """One forward pass through the HIP path with no autograd graph.  Returns
        (recon, mu, logvar) and remembers what backward_raw needs."""
        self._ensure_ready(x)
        x = x.contiguous()
        N = x.shape[0]
        hp = self._hp
        if tuple(x.shape[1:]) != (hp["in_channels"], hp["image_size"], hp["image_size"]):
            raise RuntimeError(f"expected input [N,{hp['in_channels']},{hp['image_size']},{hp['image_size']}], "
                               f"got {tuple(x.shape)}")
        train = self.training
        cfg = self._cfg(N, train)
        ws = self._workspace(N)
        dev = x.device
        recon = torch.empty(N, hp["out_channels"], hp["image_size"], hp["image_size"], device=dev)
        mu = torch.empty(N, hp["latent_dims"], device=dev)
        logvar = torch.empty(N, hp["latent_dims"], device=dev)
        if eps is not None:
            eps = eps.to(device=dev, dtype=torch.float32).contiguous()
            if tuple(eps.shape) != (N, hp["latent_dims"]):
                raise RuntimeError(f"eps must be [N, latent_dims], got {tuple(eps.shape)}")
        marr, mkeep = self._mask_ptrs(keep_masks if train else None)
        _lib.call("vae_model_forward", ctypes.byref(cfg), self.flat_params.data_ptr(), self._bn_stats.data_ptr(),
                  self._wpack.data_ptr(), ws.data_ptr(), x.data_ptr(), _lib.ptr(eps), marr, recon.data_ptr(),
                  mu.data_ptr(), logvar.data_ptr(), _lib.current_stream())
        if train:
            if not cfg.batches_tracked:
                self._nbt += 1
            self.rng_step += 1
        state = dict(cfg=cfg, ws=ws, x=x, eps=eps, marr=marr, masks=mkeep, recon=recon, mu=mu, logvar=logvar, batch=N)
        self._last = state
        return recon, mu, logvar, state

    def backward_raw(self, state, d_recon, d_mu=None, d_logvar=None, kl_scale=0.0, stage=-1, out=None):
        """Gradients of all parameters into `out` (default flat_grads) for the forward pass
        described by `state`.  stage: see vae_model_backward in include/vae_hip.h."""
        out = self.flat_grads if out is None else out
        _lib.call("vae_model_backward", ctypes.byref(state["cfg"]), self.flat_params.data_ptr(), self._wpack.data_ptr(),
                  state["ws"].data_ptr(), d_recon.data_ptr(), state["recon"].data_ptr(), state["mu"].data_ptr(),
                  state["logvar"].data_ptr(), _lib.ptr(state["eps"]), state["marr"], _lib.ptr(d_mu), _lib.ptr(d_logvar),
                  float(kl_scale), out.data_ptr(), int(stage), _lib.current_stream())
        return out

    def kl_partials(self, state):
        """(device pointer, count) of the per-workgroup KL partial sums of that forward."""
        ptr = ctypes.c_void_p()
        n = ctypes.c_int()
        _lib.call("vae_model_kl_partials", ctypes.byref(state["cfg"]), state["ws"].data_ptr(), ctypes.byref(ptr),
                  ctypes.byref(n))
        return ptr.value, n.value

    # ------------------------------------------------------------------ reference API
    def forward(self, x, eps=None, keep_masks=None):
        """recons, mu, logvar = model(x)   (vae/models/vae.py:101-105).
        eps / keep_masks optionally inject the noise that the reference draws from torch's
        global generator; by default both come from the module's Philox streams."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _VAEConvFunction.apply(self, x, eps, keep_masks, *self._ordered_params())
        recon, mu, logvar, state = self.forward_raw(x, eps, keep_masks)
        self._release_workspace(state["batch"], state["ws"])
        return recon, mu, logvar

    def encode(self, x):
        """mu, logvar = model.encode(x)  (vae/models/vae.py:89-98).  Inference helper: the
        result is not attached to an autograd graph (train through forward())."""
        self._ensure_ready(x)
        x = x.contiguous()
        N = x.shape[0]
        cfg = self._cfg(N, self.training)
        ws = self._workspace(N)
        mu = torch.empty(N, self._hp["latent_dims"], device=x.device)
        logvar = torch.empty_like(mu)
        _lib.call("vae_model_encode", ctypes.byref(cfg), self.flat_params.data_ptr(), self._bn_stats.data_ptr(),
                  self._wpack.data_ptr(), ws.data_ptr(), x.data_ptr(), None, mu.data_ptr(), logvar.data_ptr(),
                  _lib.current_stream())
        if self.training:
            if not cfg.batches_tracked:
                self._nbt[:8] += 1
            self.rng_step += 1
        self._release_workspace(N, ws)
        return mu, logvar

    def reparameterize(self, mu, logvar):
        """z = mu + exp(0.5*logvar) * eps  (vae/models/vae.py:108-111), eps from the module's
        Philox stream (the draw advances rng_step)."""
        _lib.check_device(mu)
        mu, logvar = mu.contiguous().float(), logvar.contiguous().float()
        n = mu.numel()
        z = torch.empty_like(mu)
        klp = torch.empty(_lib.call("vae_reparam_grid", n), device=mu.device)
        _lib.call("vae_reparam_kl_fwd", mu.data_ptr(), logvar.data_ptr(), None, z.data_ptr(), None, None,
                  klp.data_ptr(), n, int(self.rng_seed), int(self.rng_step) & 0xFFFFFFFF, int(self.rank),
                  _lib.current_stream())
        self.rng_step += 1
        return z

    def decode(self, z):
        """recons = model.decode(z)  (vae/models/vae.py:114-122).  Inference helper (no autograd)."""
        self._ensure_ready(z)
        z = z.contiguous()
        N = z.shape[0]
        hp = self._hp
        if tuple(z.shape) != (N, hp["latent_dims"]):
            raise RuntimeError(f"z must be [N,{hp['latent_dims']}], got {tuple(z.shape)}")
        cfg = self._cfg(N, self.training)
        ws = self._workspace(N)
        recon = torch.empty(N, hp["out_channels"], hp["image_size"], hp["image_size"], device=z.device)
        _lib.call("vae_model_decode", ctypes.byref(cfg), self.flat_params.data_ptr(), self._bn_stats.data_ptr(),
                  self._wpack.data_ptr(), ws.data_ptr(), z.data_ptr(), recon.data_ptr(), _lib.current_stream())
        if self.training:
            self._nbt[8:] += 1
        self._release_workspace(N, ws)
        return recon


class _VAEConvFunction(torch.autograd.Function):
    """autograd bridge: the whole module is one node, its backward one C call."""

    @staticmethod
    def forward(ctx, model, x, eps, keep_masks, *params):
        if x.requires_grad:
            raise RuntimeError("gradients with respect to the input images are not computed by this path")
        recon, mu, logvar, state = model.forward_raw(x, eps, keep_masks)
        ctx.model, ctx.state = model, state
        return recon, mu, logvar

    @staticmethod
    def backward(ctx, d_recon, d_mu, d_logvar):
        model, state = ctx.model, ctx.state
        if state is None:
            raise RuntimeError("backward through VAEConv a second time: the activations were already released")
        if d_recon is None:
            d_recon = torch.zeros_like(state["recon"])
        d_recon = d_recon.contiguous()
        d_mu = None if d_mu is None else d_mu.contiguous()
        d_logvar = None if d_logvar is None else d_logvar.contiguous()
        # fresh buffer: autograd may keep (or accumulate into) the tensors returned here.  Zeroed:
        # the backward leaves the identically-zero gradients (biases in front of a BatchNorm) untouched
        out = torch.zeros_like(model.flat_grads)
        model.backward_raw(state, d_recon, d_mu, d_logvar, 0.0, -1, out)
        model._release_workspace(state["batch"], state["ws"])
        ctx.state = None
        return (None, None, None, None, *model.grad_views(out))
