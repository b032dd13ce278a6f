"""Where the data-parallel step's extra time goes, as far as one GPU can show it: the plain fused step, the staged
backward with the exchange skipped, and the exchange forced on over a one-rank RCCL group (the mean over one rank
is the identity).  Prints one JSON line; run on the GPU box:  python tools_dp_overhead.py [--batch 512]"""
import argparse
import json
import socket
import time

import torch
import torch.distributed as dist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=100)
    a = ap.parse_args()
    from vae_amd.dp import BucketedReducer, stage_ranges
    from vae_amd.optim import FlatAdamW
    from vae_amd.train import Config, FusedStep
    from vae_amd.vae import VAEConv
    torch.cuda.set_device(0)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    conf = Config(kl_weight=1.0, percep_weight=0.0, recon_loss="l1", lr=1e-4)
    xs = [torch.rand(a.batch, 3, 32, 32, device="cuda") for _ in range(8)]
    out = {"batch": a.batch, "steps": a.steps}
    for mode in ("plain", "staged_no_exchange", "exchange_16MB", "exchange_one_bucket_per_stage", "exchange_from_main_stream",
                 "exchange_main_stream_two_collectives"):
        torch.manual_seed(1337)
        m = VAEConv(3, 3, 128, 0.1, 32).cuda().train()
        red = None
        if mode != "plain":
            red = BucketedReducer(m.flat_grads, stage_ranges(m._offsets, m.flat_grads.numel()),
                                  bucket_bytes=(1 << 30) if mode.endswith("per_stage") else (16 << 20),
                                  side_stream=not (mode.endswith("main_stream") or mode.endswith("two_collectives")),
                                  merge_first=mode.endswith("two_collectives"))
            if mode.startswith("exchange"):
                red.world = 2
        fs = FusedStep(m, FlatAdamW(m, lr=1e-4), conf, red)
        for i in range(20):
            fs(xs[i % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            fs(xs[i % 8])
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        out[mode] = {"ms_per_step": 1e3 * (time.perf_counter() - t0) / a.steps, "host_enqueue_ms": 1e3 * t_host / a.steps}
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
