"""Generates tests/golden/*.npz by running the REFERENCE classes themselves.

Runs only in the build container, where /root/reference is mounted read-only.
The reference's Python never enters this repository: the fixtures hold inputs,
outputs, losses and gradient digests only.  Modules the reference imports at
file top but never uses on this path (lpips, wandb, torchvision, omegaconf,
torchinfo) are absent from the image and are registered as empty stubs.

    python oracle/make_fixtures.py            # rewrites tests/golden/
"""
import os
import sys
import types
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import vae_oracle as O  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    stub("lpips")
    stub("wandb")
    tv = stub("torchvision")
    tv.utils = stub("torchvision.utils")
    tv.transforms = stub("torchvision.transforms")
    tv.datasets = stub("torchvision.datasets")
    stub("omegaconf", OmegaConf=object)
    stub("torchinfo", summary=lambda *a, **k: None)
    sys.path.insert(0, REF)
    import vae.models.vae as ref_vae
    import vae.training.train as ref_train
    return ref_vae, ref_train


LOSS_FNS = {
    "l1": F.l1_loss,
    "mse": F.mse_loss,
    "bce": F.binary_cross_entropy,
    "bce_logits": lambda i, t, reduction: F.binary_cross_entropy_with_logits(
        i.clamp(-10, 10), t, reduction=reduction),
}


def zero_lpips(target, inp, normalize=True):
    return torch.zeros(inp.size(0), 1, 1, 1)


def ref_forward_injected(model, x, eps):
    """forward with eps injected: identical expression to vae.py:109-111."""
    mu, logvar = model.encode(x)
    z = mu + torch.exp(0.5 * logvar) * eps
    return model.decode(z), mu, logvar


def build_ref_model(ref_vae, cfg, seed):
    torch.manual_seed(seed)
    m = ref_vae.VAEConv(cfg["in_channels"], cfg["out_channels"], cfg["latent_dims"],
                        cfg["p_dropout"], cfg["image_size"])
    init_sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd = O.perturb_norm_params({k: v.clone() for k, v in m.state_dict().items()}, seed)
    m.load_state_dict(sd)
    return m, init_sd


def put_digests(out, prefix, tensors, nsample=256, full_below=20000):
    for k, t in tensors.items():
        d, s = O.tensor_digest(t, nsample)
        out[f"{prefix}/{k}/digest"] = d
        out[f"{prefix}/{k}/sample"] = s
        if t.numel() <= full_below:
            out[f"{prefix}/{k}/full"] = t.detach().float().numpy().copy()


def case_forward_backward(ref_vae, ref_train, name, cfg, n, seed, train=True, kinds=O.RECON_KINDS,
                          kl_weight=1.0):
    model, init_sd = build_ref_model(ref_vae, cfg, seed)
    x, eps = O.make_inputs(n, cfg["in_channels"], cfg["image_size"], cfg["latent_dims"], seed)
    out = {"cfg/in_channels": cfg["in_channels"], "cfg/out_channels": cfg["out_channels"],
           "cfg/latent_dims": cfg["latent_dims"], "cfg/p_dropout": cfg["p_dropout"],
           "cfg/image_size": cfg["image_size"], "cfg/n": n, "cfg/seed": seed,
           "cfg/train": int(train), "cfg/kl_weight": kl_weight}
    put_digests(out, "init", init_sd, nsample=16, full_below=0)
    masks = []
    hooks = []
    if train and cfg["p_dropout"] > 0:
        def hook(mod, inp, outp):
            # keep-mask; where the input is 0 the mask is irrelevant -> 1
            masks.append(((outp != 0) | (inp[0] == 0)).to(torch.uint8))
        for i in range(1, 5):
            hooks.append(getattr(model, f"conv{i}").net[6].register_forward_hook(hook))
    model.train(train)
    conf = SimpleNamespace(percep_weight=0.0, kl_weight=kl_weight)
    torch.manual_seed(seed + 3)   # seeds the reference's own dropout draw
    bn_before = {k: v.clone() for k, v in model.state_dict().items()}
    first = True
    for kind in kinds:
        model.load_state_dict(bn_before)
        model.zero_grad()
        masks.clear()
        torch.manual_seed(seed + 3)
        if train:
            recon, mu, logvar = ref_forward_injected(model, x, eps)
        else:
            with torch.no_grad():
                recon, mu, logvar = ref_forward_injected(model, x, eps)
        loss = ref_train.vae_loss(recon, x, mu, logvar, LOSS_FNS[kind], zero_lpips, conf)
        out[f"loss/{kind}"] = np.float64(loss.item())
        if first:
            out["recon"] = recon.detach().numpy().copy()
            out["mu"] = mu.detach().numpy().copy()
            out["logvar"] = logvar.detach().numpy().copy()
            if masks:
                for i, mk in enumerate(masks):
                    out[f"mask{i}/shape"] = np.array(mk.shape)
                    out[f"mask{i}/bits"] = np.packbits(mk.numpy().reshape(-1))
            if train:
                after = model.state_dict()
                put_digests(out, "stats_after",
                            {k: v for k, v in after.items() if not O.is_param(k)}, nsample=64)
        if train:
            loss.backward()
            grads = {k: p.grad for k, p in model.named_parameters()}
            put_digests(out, f"grad/{kind}", grads,
                        nsample=1024 if first else 128, full_below=20000 if first else 0)
        first = False
    for h in hooks:
        h.remove()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")


def case_train_steps(ref_vae, ref_train, name, cfg, n, seed, steps=3):
    """full reference train-step body (train.py:327-339) with eps injected, p_dropout=0."""
    model, _ = build_ref_model(ref_vae, cfg, seed)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    conf = SimpleNamespace(percep_weight=0.0, kl_weight=1.0)
    out = {"cfg/in_channels": cfg["in_channels"], "cfg/out_channels": cfg["out_channels"],
           "cfg/latent_dims": cfg["latent_dims"], "cfg/p_dropout": cfg["p_dropout"],
           "cfg/image_size": cfg["image_size"], "cfg/n": n, "cfg/seed": seed, "cfg/steps": steps}
    for s in range(steps):
        x, eps = O.make_inputs(n, cfg["in_channels"], cfg["image_size"], cfg["latent_dims"],
                               seed + 100 * s)
        recon, mu, logvar = ref_forward_injected(model, x, eps)
        opt.zero_grad()
        loss = ref_train.l1_kl_percep_loss(recon, x, mu, logvar, conf, zero_lpips)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        opt.step()
        out[f"step{s}/loss"] = np.float64(loss.item())
        out[f"step{s}/grad_norm"] = np.float64(gn.item())
    put_digests(out, "final", dict(model.state_dict()), nsample=256, full_below=20000)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")


def case_double_conv_gn(name, cin, cout, n, size, p, seed):
    """the reference's GroupNorm double-conv block itself (vae/practice/vae_conv.py:20-36) followed by the
    MaxPool2d(2) of its encoder (:51): weights, input, train-mode output, pooled output, and the gradients of
    sum(pooled * r) for a saved random r."""
    import vae.practice.vae_conv as ref_gn
    torch.manual_seed(seed)
    blk = ref_gn.DoubleConv(cin, cout, p)
    with torch.no_grad():
        for k, v in blk.state_dict().items():      # away from the (1, 0) init of the affine
            if ".1." in k or ".4." in k:
                v.add_(0.3 * torch.randn_like(v))
    blk.train()
    masks = []
    h = blk.double_conv[6].register_forward_hook(
        lambda mod, inp, outp: masks.append(((outp != 0) | (inp[0] == 0)).to(torch.uint8)))
    x = torch.randn(n, cin, size, size, requires_grad=True)
    r = torch.randn(n, cout, size // 2, size // 2)
    y = blk(x)
    pooled = F.max_pool2d(y, 2)
    (pooled * r).sum().backward()
    h.remove()
    out = {"cfg/cin": cin, "cfg/cout": cout, "cfg/n": n, "cfg/size": size, "cfg/p": p,
           "x": x.detach().numpy().copy(), "r": r.numpy().copy(), "y": y.detach().numpy().copy(),
           "pooled": pooled.detach().numpy().copy(), "grad/x": x.grad.numpy().copy(),
           "mask/shape": np.array(masks[0].shape), "mask/bits": np.packbits(masks[0].numpy().reshape(-1))}
    for k, v in blk.state_dict().items():
        out[f"sd/{k}"] = v.detach().numpy().copy()
    for k, v in blk.named_parameters():
        out[f"grad/{k}"] = v.grad.numpy().copy()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")


def case_unet(name, latent, size, n, p, seed, kl_weight=0.5, variant="256", logvar_shift=0.0):
    """the reference's U-Net VAE itself (vae/practice/vae_original.py:77-181) in train mode with eps injected
    (same expression as its reparameterize, :178-181) and its vae_loss_mse (:527-532): logits, mu, logvar, loss,
    dropout masks of the seven blocks, gradient digests.  Weights are reproduced from the seed by
    construct_state_unet + perturb_gn_params; the fixture keeps digests of them."""
    import vae.practice.vae_original as ref_unet
    torch.manual_seed(seed)
    m = getattr(ref_unet, "VAEUnet" + variant)(3, latent, p, size)
    sd = O.perturb_gn_params(OrderedDict((k, v.clone()) for k, v in m.state_dict().items()), seed)
    if logvar_shift:                    # spreads logvar so that the [-5, 5] clamp of VAEUnet1024 binds on some entries
        sd["fc_logvar.bias"] = sd["fc_logvar.bias"] + logvar_shift * torch.linspace(-1, 1, latent)
    m.load_state_dict(sd)
    m.train()
    x, eps = O.make_inputs(n, 3, size, latent, seed)
    m.reparameterize = lambda mu, logvar: mu + torch.exp(0.5 * logvar) * eps
    masks, hooks = {}, []
    for bname, mod in m.named_children():
        if isinstance(mod, ref_unet.DoubleConv):
            hooks.append(mod.double_conv[6].register_forward_hook(
                lambda _m, inp, outp, bname=bname: masks.__setitem__(bname, ((outp != 0) | (inp[0] == 0)).to(torch.uint8))))
    torch.manual_seed(seed + 7)          # the dropout draws of this forward
    out, mu, logvar = m(x)
    loss = ref_unet.vae_loss_mse(x, out, mu, logvar, kl_weight)
    loss.backward()
    for h in hooks:
        h.remove()
    res = {"cfg/latent": latent, "cfg/size": size, "cfg/n": n, "cfg/p": p, "cfg/seed": seed, "cfg/kl_weight": kl_weight,
           "cfg/variant": variant, "cfg/logvar_shift": logvar_shift,
           "out": out.detach().numpy().copy(), "mu": mu.detach().numpy().copy(), "logvar": logvar.detach().numpy().copy(),
           "loss": np.float64(loss.item())}
    for k, v in masks.items():
        res[f"mask/{k}/shape"] = np.array(v.shape)
        res[f"mask/{k}/bits"] = np.packbits(v.numpy().reshape(-1))
    put_digests(res, "sd", sd)
    put_digests(res, "grad", {k: v.grad for k, v in m.named_parameters()})
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **res)
    print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")


def main():
    os.makedirs(OUT, exist_ok=True)
    ref_vae, ref_train = import_reference()
    if "--unet" in sys.argv:        # only the U-Net fixtures
        case_unet("unet256_s16", 64, 16, 2, 0.1, 1360)
        case_unet("unet256_s32_nodrop", 128, 32, 2, 0.0, 1361, kl_weight=1.0)
        case_unet("unet1024_s32", 64, 32, 2, 0.1, 1362, kl_weight=0.5, variant="1024", logvar_shift=8.0)
        return
    if "--gn" in sys.argv:          # only the GroupNorm block fixtures (the others stay as committed)
        case_double_conv_gn("gn_block", 8, 64, 3, 8, 0.0, 1350)
        case_double_conv_gn("gn_block_dropout", 8, 64, 2, 8, 0.25, 1351)
        return
    A = dict(in_channels=3, out_channels=3, latent_dims=128, p_dropout=0.0, image_size=32)
    Ad = dict(A, p_dropout=0.1)
    B = dict(in_channels=3, out_channels=3, latent_dims=256, p_dropout=0.0, image_size=64)
    C = dict(in_channels=3, out_channels=3, latent_dims=512, p_dropout=0.0, image_size=32)
    case_forward_backward(ref_vae, ref_train, "a32_train", A, 4, 1337)
    case_forward_backward(ref_vae, ref_train, "a32_dropout", Ad, 4, 1338, kinds=("l1",))
    case_forward_backward(ref_vae, ref_train, "a32_eval", Ad, 4, 1339, train=False)
    case_forward_backward(ref_vae, ref_train, "b64_train", B, 2, 1340, kinds=("l1", "bce"))
    case_forward_backward(ref_vae, ref_train, "c32_beta4", C, 2, 1341, kinds=("mse",), kl_weight=4.0)
    case_train_steps(ref_vae, ref_train, "a32_steps", A, 4, 1342, steps=3)


if __name__ == "__main__":
    main()
