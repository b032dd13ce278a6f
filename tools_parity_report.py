"""Per-tensor parity numbers of the HIP path, written to a JSON file (run on the GPU box; the result is committed under
profiles/).  For every golden case and for the full-size configurations it records, per output / gradient tensor,
relative L2 error, cosine and norm ratio against
  * the fp32 oracle (pinned to the reference at 1e-5 by tests/test_oracle_golden.py), and
  * the bf16-faithful oracle (oracle.forward_bf16: the kernels' own rounding points, fp32 summation order differs).

    python tools_parity_report.py [out.json]
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

DEV = "cuda"


def metrics(got, ref):
    g, r = got.detach().double().flatten().cpu(), ref.detach().double().flatten().cpu()
    nr = float(r.norm())
    return {"rel_l2": float((g - r).norm() / max(nr, 1e-30)), "cos": float(g @ r / max(float(g.norm()) * nr, 1e-30)),
            "norm_ratio": float(g.norm() / max(nr, 1e-30)), "numel": g.numel()}


def philox_inputs(N, L, S, p, seed, step, rank=0):
    eps = torch.from_numpy(O.philox_normal(N * L, seed, O.STREAM_EPS, step, rank)).view(N, L)
    masks = None
    if p > 0:
        masks = []
        for b, (C, H) in enumerate([(64, S), (128, S // 2), (256, S // 4), (256, S // 8)]):
            k = O.philox_keep_mask(N * H * H * C, p, seed, b, step, rank)
            masks.append(torch.from_numpy(k).view(N, H, H, C).permute(0, 3, 1, 2))
    return eps, masks


def run_case(name, m, sd, x, eps, cpu_masks, p, kl_weight, with_bf16=True):
    from vae_amd.train import Config, vae_loss
    masks = None if cpu_masks is None else [t.permute(0, 2, 3, 1).contiguous().to(DEV) for t in cpu_masks]
    m.zero_grad()
    recon, mu, logvar = m(x.to(DEV), eps.to(DEV), masks)
    loss = vae_loss(recon, x.to(DEV), mu, logvar, "l1", None, Config(kl_weight=kl_weight, percep_weight=0.0))
    loss.backward()
    refs = {"fp32": O.loss_and_grads(x, sd, eps, "l1", kl_weight, True, p, cpu_masks)}
    if with_bf16:
        refs["bf16"] = O.loss_and_grads_bf16(x, sd, eps, "l1", kl_weight, p, cpu_masks)
    out = {}
    for tag, (r_loss, r_recon, r_mu, r_lv, r_grads, _) in refs.items():
        rec = {"loss_rel": abs(float(loss) - float(r_loss)) / abs(float(r_loss)), "recon": metrics(recon, r_recon),
               "mu": metrics(mu, r_mu), "logvar": metrics(logvar, r_lv), "grads": {}}
        dot = ng = nr = 0.0
        for k, prm in m.named_parameters():
            if zero_grad_bias(k):
                continue
            rec["grads"][k] = metrics(prm.grad, r_grads[k])
            g, r = prm.grad.detach().double().flatten().cpu(), r_grads[k].double().flatten()
            dot, ng, nr = dot + float(g @ r), ng + float(g @ g), nr + float(r @ r)
        rec["all_grads"] = {"cos": dot / (ng * nr) ** 0.5, "norm_ratio": (ng / nr) ** 0.5}
        worst = max(rec["grads"].items(), key=lambda kv: kv[1]["rel_l2"])
        rec["worst_grad"] = {"name": worst[0], **worst[1]}
        out[tag] = rec
    print(name, {t: (round(r["recon"]["rel_l2"], 5), round(r["mu"]["rel_l2"], 5), round(r["logvar"]["rel_l2"], 5),
                     r["worst_grad"]["name"], round(r["worst_grad"]["rel_l2"], 4), round(r["all_grads"]["cos"], 5))
                 for t, r in out.items()}, flush=True)
    return out


def main():
    from vae_amd.vae import VAEConv
    report = {"note": "HIP path vs the fp32 oracle (= the reference to 1e-5) and vs the bf16-faithful oracle; per tensor: "
                      "relative L2 error, cosine, norm ratio; identical inputs, weights, eps and dropout masks"}
    for name in ("a32_train", "a32_dropout", "b64_train", "c32_beta4"):
        c = Case(name)
        torch.manual_seed(c.seed)
        m = VAEConv(**c.cfg)
        m.load_state_dict(c.state())
        m.to(DEV).train()
        x, eps = c.inputs()
        report[name] = run_case(name, m, c.state(), x, eps, c.masks(), c.cfg["p_dropout"], c.kl_weight)
    # full batch of benchmark config 2 (batch 512, 32x32, latent 128, dropout 0.1), Philox-driven
    for tag, (N, L, S) in {"cfg2_full_512": (512, 128, 32), "cfg5_full_512": (512, 512, 32)}.items():
        torch.manual_seed(1337)
        m = VAEConv(3, 3, L, 0.1, S).to(DEV).train()
        m.rng_seed, m.rank, m.rng_step = 1337, 0, 5
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        x = torch.rand(N, 3, S, S, generator=torch.Generator().manual_seed(7))
        eps, masks = philox_inputs(N, L, S, 0.1, 1337, 5)
        report[tag] = run_case(tag, m, sd, x, eps, masks, 0.1, 4.0 if L == 512 else 1.0)
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
