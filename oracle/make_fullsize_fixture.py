"""Golden digests of the CPU oracle at the FULL size of benchmark config 4 (VAEConv(3,3,256,0.1,64), batch 1024,
64x64), train mode with dropout: the oracle needs minutes and ~20 GB for this on the build machine, the GPU box only
compares.  TEST INFRASTRUCTURE (like everything under oracle/).

Everything the pass consumes is reproducible from seeds, so the fixture holds outputs only:
  weights   torch.manual_seed(1337); construct_state(3, 3, 256, 0.1, 64)   (== VAEConv(...) under the same seed,
            which tests/test_gpu_model.py::test_construction_matches_reference checks against the reference's own init)
  x         torch.rand(1024, 3, 64, 64, generator=manual_seed(11))
  eps       Philox stream 4, seed 1337, step 3, rank 0   (the model's own draw at rng_step = 3)
  masks     Philox streams 0..3, same seed / step / rank
Outputs (fp32 oracle, the restatement that tests/test_oracle_golden.py pins to the reference at 1e-5):
  recon, mu, logvar: (l2 norm, sum) + 16384 strided samples each; loss/l1; per-image L1 sums.

    python oracle/make_fullsize_fixture.py        -> tests/golden/b64_full1024.npz
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vae_oracle as O  # noqa: E402

N, L, S, P, SEED, STEP, XSEED = 1024, 256, 64, 0.1, 1337, 3, 11


def main():
    t0 = time.time()
    torch.manual_seed(SEED)
    sd = O.construct_state(3, 3, L, P, S)
    x = torch.rand(N, 3, S, S, generator=torch.Generator().manual_seed(XSEED))
    eps = torch.from_numpy(O.philox_normal(N * L, SEED, O.STREAM_EPS, STEP, 0)).view(N, L)
    masks = []
    for b, (C, H) in enumerate([(64, S), (128, S // 2), (256, S // 4), (256, S // 8)]):
        k = O.philox_keep_mask(N * H * H * C, P, SEED, b, STEP, 0)
        masks.append(torch.from_numpy(k).view(N, H, H, C).permute(0, 3, 1, 2))
    print(f"inputs ready {time.time() - t0:.0f}s", flush=True)
    with torch.no_grad():
        recon, mu, logvar = O.forward(x, sd, eps, True, P, masks)
        loss = O.vae_loss(recon, x, mu, logvar, "l1", 1.0)
        per_image = (recon - x).abs().sum(dim=(1, 2, 3))
    out = {"cfg": np.array([N, L, S, SEED, STEP, XSEED], dtype=np.int64), "p_dropout": np.float64(P),
           "loss/l1": np.float64(loss.item()), "per_image_l1": per_image.numpy().astype(np.float32)}
    for name, t in (("recon", recon), ("mu", mu), ("logvar", logvar)):
        dig, smp = O.tensor_digest(t, 16384)
        out[f"{name}/digest"], out[f"{name}/sample"] = dig, smp
    path = os.path.join(ROOT, "tests", "golden", "b64_full1024.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path} in {time.time() - t0:.0f}s: loss {loss.item():.4f}", flush=True)


if __name__ == "__main__":
    main()
