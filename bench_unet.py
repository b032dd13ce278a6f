"""Throughput of the drafts' U-Net VAE (vae_amd.unet.VAEUnet256 / VAEUnet1024; reference vae/practice/vae_original.py:77-403) on
one MI355X: full training step (forward, MSE + KL loss, backward through the autograd nodes, clip_grad_norm_, AdamW)
on synthetic data resident in HBM.  A side measurement next to bench.py (which measures the headline VAEConv path).

    python bench_unet.py [--model 256|1024] [--size 32] [--batch 512] [--latent 128] [--steps 30] [--warmup 5]
"""
import argparse
import json
import time

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--latent", type=int, default=128)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="256", choices=["256", "1024"], help="VAEUnet256 or VAEUnet1024")
    ap.add_argument("--kernels", action="store_true", help="per-kernel event times from the library's profiler")
    a = ap.parse_args()
    from vae_amd import _lib
    from vae_amd import unet
    from vae_amd.unet import vae_loss_mse
    _lib.load()
    dev = torch.device("cuda", 0)
    torch.manual_seed(1337)
    m = getattr(unet, "VAEUnet" + a.model)(3, a.latent, 0.1, a.size).to(dev).train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, fused=True)
    xs = [torch.rand(a.batch, 3, a.size, a.size, device=dev) for _ in range(4)]

    def step(x):
        out, mu, logvar = m(x)
        loss = vae_loss_mse(x, out, mu, logvar, 1.0)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        return loss

    for i in range(a.warmup):
        step(xs[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(xs[i % 4])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = {"model": f"VAEUnet{a.model}(3,{a.latent},0.1,{a.size})", "batch": a.batch, "steps": a.steps,
           "ms_per_step": 1e3 * el / a.steps, "images_per_s": a.batch * a.steps / el, "final_loss": loss.item(),
           "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}
    if a.kernels:
        import ctypes
        _lib.call("vae_prof_start", 4096)
        step(xs[0])
        torch.cuda.synchronize()
        cap = 256
        names = ctypes.create_string_buffer(cap * 64)
        launches = (ctypes.c_int * cap)()
        ms = (ctypes.c_double * cap)()
        fl = (ctypes.c_double * cap)()
        by = (ctypes.c_double * cap)()
        n = _lib.call("vae_prof_stop", names, launches, ms, fl, by, cap)
        rows = [(names.raw[i * 64:(i + 1) * 64].split(b"\0")[0].decode(), launches[i], ms[i], fl[i], by[i]) for i in range(n)]
        rows.sort(key=lambda r: -r[2])
        out["kernels"] = [{"kernel": r[0], "launches": r[1], "ms": round(r[2], 4),
                           "tflops": round(r[3] / r[2] / 1e9, 1) if r[2] > 0 else 0,
                           "gbs": round(r[4] / r[2] / 1e6, 1) if r[2] > 0 else 0} for r in rows[:16]]
        out["kernel_ms_sum"] = sum(r[2] for r in rows)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
