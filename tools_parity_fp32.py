"""fp32 mode: distance of the GPU path and of the fp32 CPU oracle from the fp64 evaluation of the same function
(forward outputs and every gradient tensor), for the reference's batch-64 configuration and a batch of 512.
Run on the GPU box: python tools_parity_fp32.py [out.json]; the result is committed under profiles/."""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from oracle import vae_oracle as O            # noqa: E402  (checker only)
from vae_amd import train as T                # noqa: E402
from vae_amd.vae import VAEConv               # noqa: E402


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def case(B, p, seed):
    torch.manual_seed(seed)
    m = VAEConv(3, 3, 128, p, 32)
    sd = O.perturb_norm_params({k: v.clone() for k, v in m.state_dict().items()}, seed)
    m.load_state_dict(sd)
    x, eps = O.make_inputs(B, 3, 32, 128, seed)
    g = torch.Generator().manual_seed(3)
    masks = [(torch.rand(B, C, 32 >> i, 32 >> i, generator=g) >= p).float() for i, C in enumerate((64, 128, 256, 256))] if p > 0 else None
    o32 = O.loss_and_grads(x, sd, eps, "l1", 1.0, True, p, masks)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    o64 = O.loss_and_grads(x.double(), sd64, eps.double(), "l1", 1.0, True, p, None if masks is None else [k.double() for k in masks])
    m.cuda().train().set_precision("fp32")
    km = None if masks is None else [k.permute(0, 2, 3, 1).contiguous().to(torch.uint8).cuda() for k in masks]
    recon, mu, logvar = m(x.cuda(), eps.cuda(), km)
    loss = T.vae_loss(recon, x.cuda(), mu, logvar, F.l1_loss, None, T.Config(kl_weight=1.0, percep_weight=0.0))
    loss.backward()
    out = {"batch": B, "p_dropout": p, "seed": seed,
           "loss_rel": {"gpu_vs_fp64": abs(loss.item() - o64[0].item()) / abs(o64[0].item()),
                        "oracle32_vs_fp64": abs(o32[0].item() - o64[0].item()) / abs(o64[0].item())}}
    for name, got, i in (("recon", recon, 1), ("mu", mu, 2), ("logvar", logvar, 3)):
        out[name] = {"gpu_vs_fp64": rel(got.detach().cpu(), o64[i]), "oracle32_vs_fp64": rel(o32[i], o64[i])}
    grads = {}
    for k, prm in m.named_parameters():
        if o64[4][k].norm() < 1e-3:
            continue        # the analytically zero bias gradients
        grads[k] = {"gpu_vs_fp64": rel(prm.grad.cpu(), o64[4][k]), "oracle32_vs_fp64": rel(o32[4][k], o64[4][k]),
                    "gpu_vs_oracle32": rel(prm.grad.cpu(), o32[4][k])}
    out["grads"] = grads
    out["grad_tensors_gpu_within_2e-5_of_fp64"] = sum(v["gpu_vs_fp64"] <= 2e-5 for v in grads.values())
    out["grad_tensors_oracle32_within_2e-5_of_fp64"] = sum(v["oracle32_vs_fp64"] <= 2e-5 for v in grads.values())
    out["grad_tensors"] = len(grads)
    return out


if __name__ == "__main__":
    res = {"note": "relative L2 per tensor; fp64 = the oracle's restatement of the reference evaluated in double; a tensor "
                   "that is 1e-4..1e-2 off in ONE of the two fp32 columns sits in front of a ReLU / max-pool decision that "
                   "this fp32 evaluation rounded the other way",
           "cases": [case(64, 0.0, 11), case(64, 0.1, 11), case(512, 0.1, 5)]}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_fp32.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    for c in res["cases"]:
        print(c["batch"], c["p_dropout"], c["recon"], c["mu"], c["grad_tensors_gpu_within_2e-5_of_fp64"],
              c["grad_tensors_oracle32_within_2e-5_of_fp64"], c["grad_tensors"])
