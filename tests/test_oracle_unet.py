"""The oracle's restatement of the drafts' U-Net VAE (vae/practice/vae_original.py:77-181) against fixtures made by
running the reference class itself (oracle/make_fixtures.py --unet): construction under the seed, train-mode
forward with injected eps and the reference's dropout masks, its vae_loss_mse, and every parameter gradient."""
import os

import numpy as np
import pytest
import torch

from oracle import vae_oracle as O
from tests.golden_util import GOLDEN, Case, check_digest, rel_l2


class UnetCase(Case):
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        g = lambda k: self.z["cfg/" + k].item()
        self.latent, self.size, self.n = int(g("latent")), int(g("size")), int(g("n"))
        self.p, self.seed, self.kl_weight = float(g("p")), int(g("seed")), float(g("kl_weight"))
        self.variant = str(g("variant")) if "cfg/variant" in self.z.files else "256"
        self.logvar_shift = float(g("logvar_shift")) if "cfg/logvar_shift" in self.z.files else 0.0

    def state(self):
        torch.manual_seed(self.seed)
        sd = O.perturb_gn_params(O.construct_state_unet(self.variant, self.latent, self.p, self.size), self.seed)
        if self.logvar_shift:
            sd["fc_logvar.bias"] = sd["fc_logvar.bias"] + self.logvar_shift * torch.linspace(-1, 1, self.latent)
        return sd

    def inputs(self):
        return O.make_inputs(self.n, 3, self.size, self.latent, self.seed)

    def masks(self):
        out = {}
        for k in self.z.files:
            if k.startswith("mask/") and k.endswith("/bits"):
                b = k.split("/")[1]
                shp = tuple(int(v) for v in self.z[f"mask/{b}/shape"])
                out[b] = torch.from_numpy(np.unpackbits(self.z[k])[:int(np.prod(shp))].reshape(shp).copy())
        return out


CASES = ["unet256_s16", "unet256_s32_nodrop", "unet1024_s32"]


@pytest.mark.parametrize("name", CASES)
def test_unet_oracle_matches_reference_fixture(name):
    c = UnetCase(name)
    sd = c.state()
    assert list(sd.keys()) == list(O.unet_state_shapes(c.variant, c.latent, c.size).keys())
    for k, v in sd.items():
        check_digest(c, "sd", k, v, 1e-7)              # construction consumes the RNG exactly as the reference
    x, eps = c.inputs()
    masks = c.masks()
    assert len(masks) == len(O.UNET_VARIANTS[c.variant][0]) + len(O.UNET_VARIANTS[c.variant][1]) and all(bool(m.all()) == (c.p == 0) for m in masks.values())
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out, mu, logvar = O.unet_forward(c.variant, x, leaf, eps, True, c.p, masks)
    if c.variant == "1024":          # the clamp binds on part of the entries, not on all
        assert 0.05 < float((logvar.detach().abs() == 5).float().mean()) < 0.95
    assert rel_l2(out.detach(), c.z["out"]) <= 1e-5
    assert rel_l2(mu.detach(), c.z["mu"]) <= 1e-5
    assert rel_l2(logvar.detach(), c.z["logvar"]) <= 1e-5
    loss = O.unet_loss_mse(x, out, mu, logvar, c.kl_weight)
    assert abs(loss.item() - float(c.z["loss"])) <= 1e-5 * abs(float(c.z["loss"]))
    loss.backward()
    for k, v in leaf.items():
        check_digest(c, "grad", k, v.grad, 2e-4, atol_scale=1e-6)
