"""The steps either side of the training path: batches and checkpoints.

* ``split_indices`` / ``DeviceFeed`` stand in for the reference's ``random_split`` + ``DataLoader``
  (vae/training/train.py:156-179) over a ``ToTensor`` dataset (train.py:258-274): the whole dataset
  sits in HBM as uint8 HWC images (CIFAR-10: 150 MB, CelebA 64x64: 2.5 GB -- a fraction of one GPU's
  memory), a batch is one gather + uint8 -> fp32/255 kernel on the training stream, and no host
  worker or PCIe copy runs per step.
* ``save_checkpoint`` / ``load_checkpoint`` write and read the reference's ``best-model.pth``: a
  plain ``state_dict`` (train.py:360-367, 187-207), loaded with ``weights_only=True``.
* ``sample`` / ``reconstruct`` / ``interpolate_latents`` / ``save_image_grid`` cover train.py:58-155
  (show_reconstructions, show_samples, interpolate_latents) without matplotlib: they return tensors and write a
  PPM/PNG grid.  ``python -m vae_amd`` runs them from a checkpoint (vae_amd/__main__.py).

* ``load_cifar10`` reads the ``cifar-10-batches-py`` files that ``datasets.CIFAR10(root, train=True)`` of
  train.py:266-274 reads, into the uint8 HWC array ``DeviceFeed`` takes -- through an unpickler that can build
  nothing but dicts, lists, strings and numpy arrays.

There is no CPU path here either: ``DeviceFeed`` calls the HIP library and fails if it is absent.
"""
from __future__ import annotations

import os
from typing import Iterator, Optional, Tuple

import numpy as np
import torch

from . import _lib


# ---------------------------------------------------------------------- CIFAR-10 batch files
class _ArrayOnlyUnpickler(__import__("pickle").Unpickler):
    """The CIFAR-10 "python version" batch files are pickles of a dict {b'data': uint8 [10000, 3072], b'labels': list,
    b'batch_label', b'filenames'}.  A pickle can name any importable callable; this unpickler resolves only the three
    globals a numpy array needs plus the bytes constructor of protocol-2 pickles (and refuses everything else), so loading a file can execute nothing."""
    # (_codecs.encode: how Python 3 writes a bytes object into a protocol-2 pickle; it turns a str into bytes)
    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy", "ndarray"), ("numpy", "dtype"), ("_codecs", "encode")}

    def find_class(self, module, name):
        if (module, name) not in self._ALLOWED:
            import pickle
            raise pickle.UnpicklingError(f"refusing to load global {module}.{name}: a CIFAR-10 batch file holds only "
                                         "dicts, lists, byte strings and numpy arrays")
        if module == "_codecs":
            import _codecs
            return _codecs.encode
        if name == "_reconstruct":
            try:
                from numpy._core.multiarray import _reconstruct      # numpy >= 2
            except ImportError:
                from numpy.core.multiarray import _reconstruct
            return _reconstruct
        return getattr(np, name)


def load_cifar10(root: str, train: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """(images uint8 [N, 32, 32, 3], labels int64 [N]) from ``root/cifar-10-batches-py`` -- ``data_batch_1..5`` for
    ``train=True`` (N = 50000), ``test_batch`` otherwise -- in file order, which is the sample order of torchvision's
    ``CIFAR10`` dataset (train.py:266-274 builds it with ``download=False``; the stored rows are CHW planes and are
    turned into HWC here, as torchvision does before ``ToTensor``).  Feed the images to ``DeviceFeed``."""
    base = os.path.join(root, "cifar-10-batches-py")
    names = [f"data_batch_{i}" for i in range(1, 6)] if train else ["test_batch"]
    images, labels = [], []
    for name in names:
        path = os.path.join(base, name)
        if not os.path.isfile(path):
            raise FileNotFoundError(f"{path} not found (download_data is not supported: place the extracted "
                                    "cifar-10-batches-py directory under data_root)")
        with open(path, "rb") as f:
            entry = _ArrayOnlyUnpickler(f, encoding="latin1").load()
        data = entry["data"] if "data" in entry else entry[b"data"]
        lab = entry.get("labels", entry.get(b"labels", entry.get("fine_labels")))
        data = np.asarray(data)
        if data.dtype != np.uint8 or data.ndim != 2 or data.shape[1] != 3072 or lab is None or len(lab) != data.shape[0]:
            raise ValueError(f"{path}: not a CIFAR-10 batch (data {data.dtype} {data.shape})")
        images.append(data.reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1))
        labels.append(np.asarray(lab, dtype=np.int64))
    return np.ascontiguousarray(np.concatenate(images)), np.concatenate(labels)


def split_indices(n: int, p_train_len: float, generator: Optional[torch.Generator] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """``random_split(dataset, [train_len, n - train_len])`` of train.py:157-159 as index tensors:
    ``train_len = int(n * p_train_len)``, one ``randperm(n)`` (torch's own ``random_split`` draws
    exactly that from the same generator), first ``train_len`` entries train, the rest validate."""
    if n <= 0:
        raise ValueError(f"empty dataset (n={n})")
    if not 0.0 <= p_train_len <= 1.0:
        raise ValueError(f"p_train_len={p_train_len} out of [0, 1]")
    train_len = int(n * p_train_len)
    perm = torch.randperm(n, generator=generator)
    return perm[:train_len].clone(), perm[train_len:].clone()


class DeviceFeed:
    """A uint8 image dataset resident on the GPU and an iterator of fp32 NCHW batches in [0, 1].

    ``images``: uint8 array / tensor ``[N, H, W, C]`` (or ``[N, H, W]`` for one channel).
    ``indices``: the subset to draw from (``split_indices``), default all.
    ``shuffle``: a fresh permutation of ``indices`` per epoch (DataLoader(shuffle=True), train.py:163),
    drawn on the host from ``generator``.  The last, short batch of an epoch is kept
    (DataLoader's ``drop_last=False``).
    """

    def __init__(self, images, batch_size: int, indices: Optional[torch.Tensor] = None, shuffle: bool = False,
                 device="cuda", generator: Optional[torch.Generator] = None):
        t = torch.as_tensor(np.asarray(images) if not torch.is_tensor(images) else images)
        if t.dtype != torch.uint8:
            raise TypeError(f"DeviceFeed wants uint8 images, got {t.dtype}")
        if t.dim() == 3:
            t = t.unsqueeze(-1)
        if t.dim() != 4 or t.size(-1) not in (1, 3, 4):
            raise ValueError(f"DeviceFeed wants [N, H, W, C] with C in (1, 3, 4), got {tuple(t.shape)}")
        if batch_size <= 0:
            raise ValueError(f"batch_size={batch_size}")
        self.n, self.h, self.w, self.c = (int(v) for v in t.shape)
        self.batch_size = int(batch_size)
        self.shuffle = bool(shuffle)
        self.generator = generator
        idx = torch.arange(self.n) if indices is None else torch.as_tensor(indices, dtype=torch.int64).clone()
        if idx.numel() == 0:
            raise ValueError("DeviceFeed: empty index set")
        if int(idx.min()) < -self.n or int(idx.max()) >= self.n:
            raise IndexError(f"DeviceFeed: index out of range for {self.n} images")
        self.indices = idx
        self.device = torch.device(device)
        _lib.load()                                   # fails loudly when the HIP library is missing
        self.data = t.contiguous().to(self.device)

    def __len__(self) -> int:
        return (self.indices.numel() + self.batch_size - 1) // self.batch_size

    def batch(self, index: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """fp32 ``[len(index), C, H, W]`` = the indexed images / 255, on the current stream.  ``index`` may live
        on the host (copied) or the device."""
        idx = index.to(device=self.device, dtype=torch.int64).contiguous()
        b = int(idx.numel())
        if out is None:
            out = torch.empty(b, self.c, self.h, self.w, device=self.device, dtype=torch.float32)
        elif tuple(out.shape) != (b, self.c, self.h, self.w) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("DeviceFeed.batch: out must be a contiguous fp32 [B, C, H, W] tensor")
        _lib.call("vae_feed_u8", self.data.data_ptr(), self.n, idx.data_ptr(), out.data_ptr(), b, self.c, self.h, self.w,
                  _lib.current_stream())
        return out

    def __iter__(self) -> Iterator[torch.Tensor]:
        order = self.indices
        if self.shuffle:
            order = order[torch.randperm(order.numel(), generator=self.generator)]
        order = order.to(self.device)                 # one small copy per epoch
        for s in range(0, order.numel(), self.batch_size):
            yield self.batch(order[s:s + self.batch_size])


# ---------------------------------------------------------------------------------------- checkpoints
def save_checkpoint(model: torch.nn.Module, path: str) -> None:
    """``torch.save(model.state_dict(), path)`` (train.py:365): tensors on the CPU, reference names and shapes."""
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(sd, path)


def load_checkpoint(model: torch.nn.Module, path: str, strict: bool = True):
    """``model.load_state_dict(torch.load(path, map_location="cpu"))`` (train.py:187,207), restricted to
    tensor-only files (``weights_only=True``: nothing in the file is executed)."""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(sd, dict):
        raise TypeError(f"{path} does not hold a state_dict")
    return model.load_state_dict(sd, strict=strict)


# ---------------------------------------------------------------------------------------- sampling
@torch.no_grad()
def sample(model, n: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """``model.decode(randn(n, latent))`` in eval mode (show_samples, train.py:86-112): ``[n, C, S, S]`` in (0, 1)."""
    was_training = model.training
    model.eval()
    dev = next(model.parameters()).device
    z = torch.randn(n, model.fc_mu.weight.shape[0], device=dev, generator=generator)     # [latent, features]
    out = model.decode(z)
    model.train(was_training)
    return out


@torch.no_grad()
def reconstruct(model, inputs: torch.Tensor) -> torch.Tensor:
    """eval-mode ``model(inputs)[0]`` (show_reconstructions, train.py:58-83)."""
    was_training = model.training
    model.eval()
    recon = model(inputs)[0]
    model.train(was_training)
    return recon


@torch.no_grad()
def interpolate_latents(model, x1: torch.Tensor, x2: torch.Tensor, num_steps: int = 10) -> torch.Tensor:
    """The reference's latent walk (interpolate_latents, train.py:115-155): encode two images to their means, decode
    ``(1 - a) * mu1 + a * mu2`` for ``a`` in ``linspace(0, 1, num_steps)``.  ``x1`` / ``x2``: ``[C, S, S]`` or
    ``[1, C, S, S]`` in [0, 1].  Returns ``[num_steps, C_out, S, S]`` (all steps decoded as ONE batch)."""
    if num_steps < 2:
        raise ValueError(f"num_steps={num_steps}: need at least the two end points")
    was_training = model.training
    model.eval()
    dev = next(model.parameters()).device
    pair = torch.stack([x1.reshape(x1.shape[-3:]), x2.reshape(x2.shape[-3:])]).to(device=dev, dtype=torch.float32)
    mu, _ = model.encode(pair)
    alpha = torch.linspace(0, 1, steps=num_steps, device=dev).unsqueeze(1)
    z = (1 - alpha) * mu[0:1] + alpha * mu[1:2]
    out = model.decode(z.contiguous())
    model.train(was_training)
    return out


def save_image_grid(images: torch.Tensor, path: str, nrow: int = 8, padding: int = 2) -> None:
    """Tile ``[N, C, H, W]`` images in [0, 1] into one picture (the layout of torchvision's make_grid: ``nrow`` images
    per row, ``padding`` black pixels between them) and write it: ``.ppm`` / ``.pgm`` natively, anything else via PIL."""
    x = images.detach().float().clamp(0, 1).cpu()
    if x.dim() != 4:
        raise ValueError("save_image_grid wants [N, C, H, W]")
    n, c, h, w = x.shape
    if c == 1:
        x = x.expand(n, 3, h, w)
    elif c != 3:
        raise ValueError(f"save_image_grid: {c} channels")
    cols = min(nrow, n)
    rows = (n + cols - 1) // cols
    grid = torch.zeros(3, rows * (h + padding) + padding, cols * (w + padding) + padding)
    for i in range(n):
        r, q = divmod(i, cols)
        y0, x0 = padding + r * (h + padding), padding + q * (w + padding)
        grid[:, y0:y0 + h, x0:x0 + w] = x[i]
    arr = (grid * 255.0 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).contiguous().numpy()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if path.lower().endswith((".ppm", ".pgm")):
        with open(path, "wb") as f:
            f.write(f"P6\n{arr.shape[1]} {arr.shape[0]}\n255\n".encode())
            f.write(arr.tobytes())
    else:
        from PIL import Image
        Image.fromarray(arr).save(path)
