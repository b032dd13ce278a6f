"""Loading tests/golden/*.npz and rebuilding the exact inputs they were made from."""
import os

import numpy as np
import torch

from oracle import vae_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Case:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        g = lambda k: self.z["cfg/" + k].item()
        self.cfg = dict(in_channels=int(g("in_channels")), out_channels=int(g("out_channels")),
                        latent_dims=int(g("latent_dims")), p_dropout=float(g("p_dropout")),
                        image_size=int(g("image_size")))
        self.n = int(g("n"))
        self.seed = int(g("seed"))
        self.train = bool(g("train")) if "cfg/train" in self.z else True
        self.kl_weight = float(g("kl_weight")) if "cfg/kl_weight" in self.z else 1.0

    def init_state(self):
        """state right after construction under torch.manual_seed(seed) (oracle restatement)."""
        torch.manual_seed(self.seed)
        return O.construct_state(**self.cfg)

    def state(self):
        return O.perturb_norm_params(self.init_state(), self.seed)

    def inputs(self, seed=None):
        c = self.cfg
        return O.make_inputs(self.n, c["in_channels"], c["image_size"], c["latent_dims"],
                             self.seed if seed is None else seed)

    def masks(self):
        if "mask0/bits" not in self.z:
            return None
        out = []
        for i in range(4):
            shp = tuple(int(v) for v in self.z[f"mask{i}/shape"])
            n = int(np.prod(shp))
            out.append(torch.from_numpy(np.unpackbits(self.z[f"mask{i}/bits"])[:n].reshape(shp).copy()))
        return out

    def kinds(self):
        return [k.split("/")[1] for k in self.z.files if k.startswith("loss/")]

    def names(self, prefix):
        return sorted({k[len(prefix) + 1:].rsplit("/", 1)[0] for k in self.z.files
                       if k.startswith(prefix + "/")})


def rel_l2(a, b):
    a = torch.as_tensor(a).double().reshape(-1)
    b = torch.as_tensor(b).double().reshape(-1)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def check_digest(case, prefix, name, t, rtol, atol_scale=0.0, nsample=256):
    """compare tensor ``t`` with the stored digest: norm, strided samples, full copy if present.
    Errors are measured relative to the tensor's own L2 norm (per-tensor relative error);
    ``atol_scale`` adds an absolute floor for tensors that are analytically ~0."""
    z = case.z
    dig, smp = O.tensor_digest(t, nsample)
    ref_dig = z[f"{prefix}/{name}/digest"]
    ref_smp = z[f"{prefix}/{name}/sample"]
    assert smp.shape == ref_smp.shape, (name, smp.shape, ref_smp.shape)
    numel = t.numel()
    rms = ref_dig[0] / np.sqrt(numel)
    tol_abs = rtol * rms * 8 + atol_scale       # per-sample allowance: 8 sigma of the rms error
    assert abs(dig[0] - ref_dig[0]) <= rtol * ref_dig[0] + atol_scale * np.sqrt(numel), \
        f"{prefix}/{name}: norm {dig[0]} vs {ref_dig[0]}"
    err = np.abs(smp.astype(np.float64) - ref_smp.astype(np.float64)).max() if smp.size else 0.0
    assert err <= tol_abs, f"{prefix}/{name}: sample err {err} > {tol_abs}"
    key = f"{prefix}/{name}/full"
    if key in z.files:
        ref = torch.from_numpy(z[key])
        d = (t.detach().float().reshape(-1) - ref.reshape(-1)).double().norm().item()
        assert d <= rtol * ref_dig[0] + atol_scale * np.sqrt(numel), \
            f"{prefix}/{name}: full l2 err {d} vs norm {ref_dig[0]}"


def zero_grad_bias(name: str) -> bool:
    """Biases of convs / transposed convs that feed a train-mode BatchNorm: their gradient is
    analytically zero (the batch mean absorbs the bias), so the reference holds only rounding
    noise there (~1e-6) and a relative comparison is meaningless -> absolute floor."""
    import re
    return bool(re.fullmatch(r"conv\d\.net\.[03]\.bias|up\d\.bias", name))


def digest_rel_err(case, prefix, name, t, nsample=256):
    """relative L2 error of tensor ``t`` against the stored golden entry: exact over the full
    tensor when the fixture holds it, otherwise estimated on the strided sample; also returns
    the reference rms (to apply an absolute floor to analytically-zero tensors)."""
    z = case.z
    ref_dig = z[f"{prefix}/{name}/digest"]
    rms = ref_dig[0] / np.sqrt(max(t.numel(), 1))
    key = f"{prefix}/{name}/full"
    if key in z.files:
        ref = torch.from_numpy(z[key]).reshape(-1).double()
        got = t.detach().float().cpu().reshape(-1).double()
    else:
        _, smp = O.tensor_digest(t.detach().float().cpu(), nsample)
        ref = torch.from_numpy(z[f"{prefix}/{name}/sample"]).double()
        got = torch.from_numpy(smp).double()
    err = (got - ref).norm().item()
    return err / max(ref.norm().item(), 1e-30), err / np.sqrt(max(ref.numel(), 1)), rms
