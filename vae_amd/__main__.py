"""Command-line entry of the steps around training that the reference runs from ``load_and_test_model``
(vae/training/train.py:182-229: load ``best-model.pth``, ``show_samples``, ``interpolate_latents``) and
``show_reconstructions`` (train.py:58-84), without wandb or matplotlib: pictures are written as PNG / PPM grids.

    python -m vae_amd sample       --checkpoint best-model.pth --n 16 --out samples.png
    python -m vae_amd reconstruct  --checkpoint best-model.pth --images val.npy --n 8 --out recon.png
    python -m vae_amd interpolate  --checkpoint best-model.pth --images val.npy --steps 10 --out walk.png

``--config ENV`` picks config/{ENV}.yaml over config/base.yaml for the model's shape (image_size, latent_dims,
num_channels, p_dropout), as the reference's ``load_config`` does; ``--images`` is a uint8 ``.npy`` array
``[N, H, W, C]`` (``ToTensor`` semantics: value / 255); without it random images stand in.  ``--precision fp32``
runs the fp32 kernels.  Without ``--checkpoint`` the freshly initialised model is used (a smoke run).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import torch


def _images(args, conf, n, dev):
    c, s = int(conf.num_channels), int(conf.image_size)
    if args.images:
        arr = np.load(args.images, allow_pickle=False)
        if arr.dtype != np.uint8 or arr.ndim != 4 or arr.shape[1:] != (s, s, c):
            raise SystemExit(f"--images: expected uint8 [N, {s}, {s}, {c}], got {arr.dtype} {arr.shape}")
        if arr.shape[0] < n:
            raise SystemExit(f"--images holds {arr.shape[0]} images, {n} needed")
        g = torch.Generator().manual_seed(int(args.seed))
        pick = torch.randperm(arr.shape[0], generator=g)[:n].numpy()
        return torch.from_numpy(arr[np.sort(pick)]).permute(0, 3, 1, 2).float().div(255.0).to(dev)
    g = torch.Generator(device=dev).manual_seed(int(args.seed))
    return torch.rand(n, c, s, s, device=dev, generator=g)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m vae_amd", description=__doc__.split("\n\n")[0])
    ap.add_argument("command", choices=["sample", "reconstruct", "interpolate"])
    ap.add_argument("--checkpoint", default=None, help="a state_dict file (the reference's best-model.pth)")
    ap.add_argument("--config", default="cfg2_1gpu", help="config/{name}.yaml over config/base.yaml")
    ap.add_argument("--config-dir", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config"))
    ap.add_argument("--images", default=None, help="uint8 .npy [N, H, W, C]")
    ap.add_argument("--n", type=int, default=8, help="images to sample / reconstruct")
    ap.add_argument("--steps", type=int, default=10, help="interpolation steps")
    ap.add_argument("--out", default=None, help="output picture (.png / .ppm); default {command}.png")
    ap.add_argument("--seed", type=int, default=1337)
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16")
    args = ap.parse_args(argv)

    from .data import interpolate_latents, load_checkpoint, reconstruct, sample, save_image_grid
    from .train import load_config
    from .vae import VAEConv
    if not torch.cuda.is_available():
        raise SystemExit("python -m vae_amd needs an MI355X: there is no CPU execution path")
    conf = load_config(args.config, args.config_dir)
    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(int(args.seed))
    model = VAEConv(int(conf.num_channels), 3, int(conf.latent_dims), float(conf.p_dropout), int(conf.image_size))
    if args.checkpoint:
        load_checkpoint(model, args.checkpoint)
    model.to(dev).eval().set_precision(args.precision)
    out = args.out or f"{args.command}.png"
    if args.command == "sample":
        g = torch.Generator(device=dev).manual_seed(int(args.seed))
        pics, nrow = sample(model, args.n, generator=g), args.n
    elif args.command == "reconstruct":
        x = _images(args, conf, args.n, dev)
        pics, nrow = torch.cat([x[:, :3] if x.size(1) >= 3 else x.expand(-1, 3, -1, -1), reconstruct(model, x)]), args.n
    else:
        x = _images(args, conf, 2, dev)
        pics, nrow = interpolate_latents(model, x[0], x[1], args.steps), args.steps
    save_image_grid(pics, out, nrow=nrow)
    print(f"{args.command}: wrote {tuple(pics.shape)} to {out}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
