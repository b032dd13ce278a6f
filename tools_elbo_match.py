"""Matched-ELBO check of the bf16 path against the fp32 mode: the reference's train loop body
(vae/training/train.py:324-340) run for hundreds of steps in both precisions from the same initial weights, on the same
batches, with the same Philox streams (eps and dropout masks are functions of (seed, step, rank) only, and both modes
index the dropout stream by the same NHWC element index), followed by the reference's validation body
(train.py:346-355) on held-out batches.

The fp32 mode reproduces the fp32 reference to rounding (tests/test_gpu_fp32_mode.py), so a bf16 loss curve that stays on
the fp32 mode's curve is a bf16 run that trains like the reference.  Single bf16 gradients differ from fp32 ones by
10-40 % per tensor (rounding flips ReLU / max-pool routes); what this measures is whether that noise matters over a run.

    python tools_elbo_match.py [out.json] [steps] [batch]      # on the GPU box; the result is committed under profiles/
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DEV = "cuda"


def synthetic_images(n, size, seed, device=DEV):
    """Images with structure to learn (uniform noise has none: every model ends at the grey image): a smooth random
    field per channel (bilinear upsampling of a 4x4 grid of normals) plus a fainter 8x8 field, squashed into (0, 1)."""
    g = torch.Generator(device=device).manual_seed(seed)
    coarse = torch.randn(n, 3, 4, 4, device=device, generator=g)
    fine = torch.randn(n, 3, 8, 8, device=device, generator=g)
    up = torch.nn.functional.interpolate
    f = 1.5 * up(coarse, size=(size, size), mode="bilinear", align_corners=False) \
        + 0.5 * up(fine, size=(size, size), mode="bilinear", align_corners=False)
    return torch.sigmoid(f).contiguous()


def run_curve(precision, kind, steps, batch, n_train_batches=16, n_val_batches=4, kl_weight=1.0, lr=1e-4, latent=128,
              size=32, p_dropout=0.1, seed=1337, reducer_factory=None):
    """`steps` fused train steps, then the eval-mode validation loss; returns (per-step losses, val loss)."""
    from vae_amd.optim import FlatAdamW
    from vae_amd.train import Config, FusedStep, val_step
    from vae_amd.vae import VAEConv
    torch.manual_seed(seed)
    m = VAEConv(3, 3, latent, p_dropout, size).to(DEV).train()
    m.rng_seed, m.rank, m.rng_step = seed, 0, 0
    m.set_precision(precision)
    conf = Config(kl_weight=kl_weight, percep_weight=0.0, recon_loss=kind, lr=lr)
    data = synthetic_images((n_train_batches + n_val_batches) * batch, size, seed + 1)
    train = [data[i * batch:(i + 1) * batch] for i in range(n_train_batches)]
    val = [data[(n_train_batches + i) * batch:(n_train_batches + i + 1) * batch] for i in range(n_val_batches)]
    reducer = reducer_factory(m) if reducer_factory else None
    step = FusedStep(m, FlatAdamW(m, lr=lr), conf, reducer)
    losses = torch.empty(steps, device=DEV)
    for i in range(steps):
        losses[i] = step(train[i % n_train_batches])
    m.eval()
    m.rng_step = 1 << 20                      # the validation eps stream: the same in every run
    vl = torch.stack([val_step(m, x, conf) for x in val]).mean()
    torch.cuda.synchronize()
    return losses.cpu().double(), float(vl)


def smoothed(losses, window=20):
    """mean over consecutive windows of `window` steps (the checkpoints the curves are compared at)."""
    n = losses.numel() // window
    return losses[: n * window].view(n, window).mean(1)


def compare(a, b, window=20):
    sa, sb = smoothed(a, window), smoothed(b, window)
    rel = ((sa - sb).abs() / sb.abs())
    return {"checkpoints": sa.numel(), "worst_rel_diff": float(rel.max()), "mean_rel_diff": float(rel.mean()),
            "curve_a": [round(float(v), 4) for v in sa], "curve_b": [round(float(v), 4) for v in sb]}


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/elbo_match.json"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    from vae_amd import _lib
    _lib.load()
    rec = {"what": "bf16 path vs fp32 mode: same init, same batches, same Philox streams; 20-step-smoothed training loss "
                   "at every checkpoint and the eval-mode validation loss afterwards",
           "config": {"model": "VAEConv(3,3,128,0.1,32)", "batch": batch, "steps": steps, "lr": 1e-4, "kl_weight": 1.0,
                      "data": "synthetic smooth fields (tools_elbo_match.synthetic_images), 16 training + 4 validation batches"},
           "runs": {}}
    for kind in ("l1", "bce", "mse"):
        lb, vb = run_curve("bf16", kind, steps, batch)
        lf, vf = run_curve("fp32", kind, steps, batch)
        r = compare(lb, lf)
        r.update(val_bf16=vb, val_fp32=vf, val_rel_diff=abs(vb - vf) / abs(vf), first_loss=float(lf[0]), last_loss=float(lf[-1]))
        rec["runs"][kind] = r
        print(kind, "worst checkpoint", r["worst_rel_diff"], "val", vb, vf, flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, "w") as f:
        json.dump(rec, f, indent=1)


if __name__ == "__main__":
    main()
