"""How far apart two bf16 pipelines of this depth can be when NOTHING differs but the fp32 summation order (CPU only).

The bf16-faithful oracle (oracle.forward_bf16: bf16 operands and stored activations, fp32 accumulation -- the kernels'
rounding points) is evaluated twice on a golden case: as it stands, and with the 64 channels between the first two
convolutions renumbered (weights, BatchNorm parameters and the next conv's input channels permuted together).  The
two networks are the same function with bit-identical bf16 operands; only the order in which conv1.net.3 adds its 576
products changes.  In fp32 the results agree to 2e-6.  In the bf16 pipeline a difference of one fp32 ulp flips a
bf16 rounding here and there, each flip is a 2^-8 relative change, and every later rounding turns a perturbation d
into ~sqrt(d * 2^-8): within a few layers the two runs sit a bf16 rounding apart everywhere, and BatchNorm's mean
subtraction magnifies that.  The numbers below are therefore the floor for comparing the HIP path with the bf16 oracle.

Also recorded: the distance of the bf16 oracle from the fp32 oracle, and the same with the stored pre-BatchNorm tensor
re-centred per channel before rounding (the best a shift of the stored tensor could do).

    python tools_bf16_noise_floor.py [out.json]  (the result is committed under profiles/)
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import vae_oracle as O  # noqa: E402
from golden_util import Case, zero_grad_bias  # noqa: E402

PERMUTED = ("conv1.net.0.weight", "conv1.net.0.bias", "conv1.net.1.weight", "conv1.net.1.bias", "conv1.net.1.running_mean",
            "conv1.net.1.running_var")


def renumbered(sd, seed=0):
    perm = torch.randperm(64, generator=torch.Generator().manual_seed(seed))
    sd = {k: v.clone() for k, v in sd.items()}
    for k in PERMUTED:
        sd[k] = sd[k][perm]
    sd["conv1.net.3.weight"] = sd["conv1.net.3.weight"][:, perm]
    return sd, torch.argsort(perm)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def reorder_experiment(name):
    c = Case(name)
    x, eps = c.inputs()
    sd = c.state()
    sdp, inv = renumbered(sd)
    out = {}
    for tag, fn in (("bf16", lambda s: O.loss_and_grads_bf16(x, s, eps, "l1", 1.0, 0.0, None)),
                    ("fp32", lambda s: O.loss_and_grads(x, s, eps, "l1", 1.0, True, 0.0, None))):
        l0, r0, mu0, lv0, g0, _ = fn(sd)
        l1, r1, mu1, lv1, g1, _ = fn(sdp)
        grads = {}
        for k in g0:
            if zero_grad_bias(k):
                continue
            a = g1[k]
            if k in PERMUTED:
                a = a[inv]
            if k == "conv1.net.3.weight":
                a = a[:, inv]
            grads[k] = rel(a, g0[k])
        worst = max(grads, key=grads.get)
        out[tag] = {"loss_rel": abs(float(l1) - float(l0)) / abs(float(l0)), "recon": rel(r1, r0), "mu": rel(mu1, mu0),
                    "logvar": rel(lv1, lv0), "worst_grad": {"name": worst, "rel_l2": grads[worst]}}
    return out


def shift_experiment(name):
    c = Case(name)
    x, eps = c.inputs()
    sd = c.state()
    with torch.no_grad():
        f = O.forward(x, sd, eps, True, 0.0, None)
        b = O.forward_bf16(x, sd, eps, True, 0.0, None)
        orig = O._bn_bf16

        def centred(y, gamma, beta, rm, rv, train, stats_out, prefix):
            mean = y.mean(dim=(0, 2, 3))
            var = y.var(dim=(0, 2, 3), unbiased=False)
            scale = gamma * torch.rsqrt(var + O.BN_EPS)
            return O._bf(y - mean[None, :, None, None]) * scale[None, :, None, None] + beta[None, :, None, None]
        O._bn_bf16 = centred
        try:
            s = O.forward_bf16(x, sd, eps, True, 0.0, None)
        finally:
            O._bn_bf16 = orig
    keys = ("recon", "mu", "logvar")
    return {"bf16_vs_fp32": {k: rel(b[i], f[i]) for i, k in enumerate(keys)},
            "bf16_recentred_vs_fp32": {k: rel(s[i], f[i]) for i, k in enumerate(keys)}}


def main():
    rep = {"note": __doc__.split("\n\n")[0]}
    for name in ("a32_train", "b64_train"):
        rep[name] = {"same_function_other_summation_order": reorder_experiment(name), "stored_tensor_recentred": shift_experiment(name)}
        print(name, json.dumps(rep[name]))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "bf16_noise_floor.json")
    with open(path, "w") as f:
        json.dump(rep, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
