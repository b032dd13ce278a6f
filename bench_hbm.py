"""Scaled micro-benchmarks of the HBM-bound kernels of the training path (loss, reparameterisation, BatchNorm /
activation passes, the 1x1 head, clip + AdamW): the tensors of the benchmark configurations are too small to
say anything about a kernel's streaming rate (the reparameterisation of config 2 moves 0.8 MB), so each kernel
is run here on inputs that move at least 256 MB per launch, and its ALGORITHMIC bytes (every input read once,
every output written once) are divided by the launch time.  Prints one JSON line; peak = 8000 GB/s.

    python bench_hbm.py            # on cuda:0
"""
import json
import sys

import torch

PEAK_GBS = 8000.0


def timeit(fn, iters=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e-3 / iters


def main():
    from vae_amd import _lib as L
    L.load()
    dev = "cuda"
    s = L.current_stream
    rows = []

    def add(name, nbytes, fn):
        t = timeit(fn)
        rows.append({"kernel": name, "bytes_per_launch": nbytes, "us": 1e6 * t, "gbs": nbytes / t / 1e9,
                     "frac": nbytes / t / 1e9 / PEAK_GBS})

    # ---- reconstruction loss, forward + gradient in one pass: recon, target in; grad out (fp32)
    n = 32 << 20
    recon, target = torch.rand(n, device=dev), torch.rand(n, device=dev)
    grad = torch.empty(n, device=dev)
    partial = torch.empty(L.call("vae_recon_loss_grid", n), device=dev)
    for kind, nm in ((0, "l1"), (1, "mse"), (2, "bce")):
        add(f"recon_loss_fwd_bwd({nm})", 12.0 * n, lambda kind=kind: L.call(
            "vae_recon_loss_fwd_bwd", recon.data_ptr(), target.data_ptr(), grad.data_ptr(), partial.data_ptr(), n, kind,
            1.0 / 512, s()))
    del recon, target, grad

    # ---- reparameterisation + KL forward: mu, logvar in; z (fp32 + bf16) out; eps from Philox
    n = 32 << 20
    mu, lv = torch.randn(n, device=dev), torch.randn(n, device=dev) * 0.1
    z, zb = torch.empty(n, device=dev), torch.empty(n, device=dev, dtype=torch.bfloat16)
    klp = torch.empty(L.call("vae_reparam_grid", n), device=dev)
    add("reparam_kl_fwd", 14.0 * n, lambda: L.call("vae_reparam_kl_fwd", mu.data_ptr(), lv.data_ptr(), None, z.data_ptr(),
                                                    zb.data_ptr(), None, klp.data_ptr(), n, 1337, 0, 0, s()))
    del mu, lv, z, zb

    # ---- BatchNorm + ReLU (+ dropout + 2x2 max-pool) forward and backward on a 64-channel 32x32 tensor of 2048 images
    N, H, C = 2048, 32, 64
    e = N * H * H * C
    y = (torch.randn(N, H, H, C, device=dev) * 1.5).to(torch.bfloat16)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) - 0.5
    mean, rstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    out = torch.empty(N, H, H, C, device=dev, dtype=torch.bfloat16)
    add("bn_act_fwd", 4.0 * e, lambda: L.call("vae_bn_act_fwd", y.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(),
                                              N, H, H, C, 0, 0.0, None, 0, 0, 0, 0, s()))
    route = torch.empty(N, H // 2, H // 2, C, device=dev, dtype=torch.uint8)
    outp = torch.empty(N, H // 2, H // 2, C, device=dev, dtype=torch.bfloat16)
    add("bn_act_fwd<pool, dropout 0.1>", 2.0 * e * 1.25 + e / 4, lambda: L.call(
        "vae_bn_act_fwd_route", y.data_ptr(), sc.data_ptr(), sh.data_ptr(), outp.data_ptr(), route.data_ptr(), N, H, H, C, 1,
        0.1, None, 1337, 0, 0, 0, s()))
    nb = L.call("vae_bn_act_bwd_blocks", N, H, H, C, 0)
    part = torch.empty(nb, 2 * C, device=dev)
    sums = torch.empty(2 * C, device=dev)
    g = torch.randn(N, H, H, C, device=dev).to(torch.bfloat16)
    dy = torch.empty_like(y)
    add("bn_act_bwd(reduce+apply)", 2.0 * e * 5, lambda: L.call(
        "vae_bn_act_bwd", g.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
        part.data_ptr(), sums.data_ptr(), dy.data_ptr(), N, H, H, C, 0, 0.0, None, 0, 0, 0, 0, s()))
    gp = torch.randn(N, H // 2, H // 2, C, device=dev).to(torch.bfloat16)
    nbp = L.call("vae_bn_act_bwd_blocks", N, H, H, C, 1)
    partp = torch.empty(nbp, 2 * C, device=dev)
    add("bn_act_bwd(reduce+apply)<pool, routed>", 2.0 * e * (2 + 0.5 + 1) + 2 * e / 4, lambda: L.call(
        "vae_bn_act_bwd_route", gp.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
        partp.data_ptr(), sums.data_ptr(), dy.data_ptr(), route.data_ptr(), N, H, H, C, 1, 0.1, None, 1337, 0, 0, 0, s()))

    # ---- first conv's BatchNorm backward + weight gradient in one pass: gin, y and the 8-channel input read once
    x8 = torch.rand(N, H, H, 8, device=dev).to(torch.bfloat16)
    scratch = torch.empty(L.call("vae_bn_bwd_conv1_scratch_floats", N, H, H), device=dev)
    dw = torch.empty(64 * 27, device=dev)
    add("bn_bwd_conv1 (BatchNorm backward + conv weight gradient, one pass)", N * H * H * (2.0 * 2 * C + 16.0), lambda: L.call(
        "vae_bn_bwd_conv1", g.data_ptr(), y.data_ptr(), x8.data_ptr(), sc.data_ptr(), sh.data_ptr(), mean.data_ptr(),
        rstd.data_ptr(), scratch.data_ptr(), sums.data_ptr(), dw.data_ptr(), N, H, H, 3, s()))
    del x8, scratch

    # ---- GroupNorm(8) + ReLU (+ dropout + pool): one workgroup per image; algorithmic bytes count y ONCE per pass
    # (the second read of an image comes from L2), gin and the output once
    mr = torch.empty(N, 8, 2, device=dev)
    add("gn_act_fwd", 4.0 * e, lambda: L.call("vae_gn_act_fwd", y.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(),
                                              mr.data_ptr(), N, H, H, C, 8, 1e-5, 0, 0.0, None, 0, 0, 0, 0, s()))
    add("gn_act_fwd<pool, dropout 0.1>", 2.0 * e * 1.25, lambda: L.call(
        "vae_gn_act_fwd", y.data_ptr(), sc.data_ptr(), sh.data_ptr(), outp.data_ptr(), mr.data_ptr(), N, H, H, C, 8, 1e-5, 1,
        0.1, None, 1337, 0, 0, 0, s()))
    cpart, csums = torch.empty(N, 3, C, device=dev), torch.empty(3 * C, device=dev)
    add("gn_act_bwd", 2.0 * e * 3, lambda: L.call(
        "vae_gn_act_bwd", g.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr(), mr.data_ptr(), cpart.data_ptr(),
        csums.data_ptr(), dy.data_ptr(), N, H, H, C, 8, 0, 0.0, None, 0, 0, 0, 0, s()))
    add("gn_act_bwd<pool, dropout 0.1>", 2.0 * e * 2.25, lambda: L.call(
        "vae_gn_act_bwd", gp.data_ptr(), y.data_ptr(), sc.data_ptr(), sh.data_ptr(), mr.data_ptr(), cpart.data_ptr(),
        csums.data_ptr(), dy.data_ptr(), N, H, H, C, 8, 1, 0.1, None, 1337, 0, 0, 0, s()))

    # ---- bn1 + ReLU + 1x1 head + sigmoid: reads the 64-channel tensor, writes 3 fp32 planes
    w, b = torch.randn(3, 64, device=dev) * 0.1, torch.zeros(3, device=dev)
    rec = torch.empty(N, 3, H, H, device=dev)
    add("head_fwd", N * H * H * (128.0 + 12.0), lambda: L.call("vae_head_fwd", y.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                                                               w.data_ptr(), b.data_ptr(), rec.data_ptr(), N, H, H, 3, s()))
    del y, out, outp, g, gp, dy, route

    # ---- global-norm clip + AdamW: grad read twice (norm, update); param, m, v read and written
    n = 16 << 20
    p_, g_ = torch.randn(n, device=dev), torch.randn(n, device=dev) * 1e-3
    m_, v_ = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    part = torch.empty(L.call("vae_adamw_grid", n), device=dev)
    norm = torch.zeros(1, device=dev)
    add("sumsq + adamw_clip", 32.0 * n, lambda: L.call(
        "vae_adamw_clip_step", p_.data_ptr(), g_.data_ptr(), m_.data_ptr(), v_.data_ptr(), n, part.data_ptr(), 1.0, 1.0, 1e-4,
        0.9, 0.999, 1e-8, 0.01, 1, norm.data_ptr(), s()))

    print(json.dumps({"metric": "algorithmic HBM bandwidth of the memory-bound kernels at scaled sizes", "unit": "GB/s",
                      "peak": PEAK_GBS, "device": torch.cuda.get_device_name(0), "kernels": rows}))


if __name__ == "__main__":
    sys.exit(main())
