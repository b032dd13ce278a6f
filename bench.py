"""Headline benchmark: training images/sec of the conv-VAE hot path on N MI355X GPUs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2_1gpu]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started plainly with --gpus N > 1 (no WORLD_SIZE in the environment) the process makes no GPU call: it starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a child, one rank per GPU, passes rank 0's
JSON line through and exits with the child's status.

A step is one full train step on one synthetic batch already resident in HBM: forward,
ELBO loss (+ its gradient), backward, global-norm clip, AdamW, bf16 weight repack -- the
reference's loop body (vae/training/train.py:327-339) -- at BASELINE config 2
(VAEConv(3,3,128,0.1,32), batch 512 per GPU, bf16 matrix-core operands).  With N > 1 every rank
runs the same per-GPU batch (weak scaling) and gradients are averaged by a bucketed RCCL
all-reduce overlapped with the backward pass.

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline     : the kernel with the largest share of a step, timed with HIP events on its
                 own stream in a separate instrumented pass of the same workload
  kernels      : the same measurement for every instrumented kernel
  cpu_baseline : the CPU oracle's train step (config 1: batch 64, fp32) on this host's cores
  secondary    : (one GPU, default config only) the other single-GPU workloads of the benchmark configs, each a short
                 run of its own after the headline's timed region: config 4 training, config 5 training, config 5
                 z -> image decoding, config 1 in the fp32 mode -- {config, dtype, value, ms_per_step, roofline}
"""
import argparse
import ctypes
import json
import os
import sys
import time

# RCCL shares device memory between the ranks through dmabuf handles; the legacy IPC mode fails on this driver
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_TFLOPS_BF16 = 2500.0     # dense bf16 MFMA peak of one MI355X
PEAK_HBM_GBS = 8000.0         # HBM3E peak of one MI355X
PEAK_TFLOPS_FP32 = 157.3      # fp32 vector / matrix peak of one MI355X (the fp32 mode's kernels)


def train_flops_per_image(in_ch, latent, S):
    """algorithmic conv/linear FLOPs of one train step per image: 3 x 2 x forward MACs, minus the
    input-gradient of the first conv (the images need no gradient)."""
    enc = [(in_ch, 64, S), (64, 64, S), (64, 128, S // 2), (128, 128, S // 2), (128, 256, S // 4),
           (256, 256, S // 4), (256, 256, S // 8), (256, 256, S // 8)]
    macs = sum(ci * co * 9 * h * h for ci, co, h in enc)
    F = 256 * (S // 8) ** 2
    macs += 2 * F * latent + latent * F
    macs += 256 * 128 * 4 * (S // 8) ** 2 + 128 * 64 * 4 * (S // 4) ** 2 + 64 * 64 * 4 * (S // 2) ** 2
    macs += 64 * 3 * S * S
    first = in_ch * 64 * 9 * S * S
    return 6 * macs - 2 * first


def family(name):
    """kernel family = the name without its template arguments (conv3_patch_tn_kernel<64> and <128> are one family)."""
    return name.split("<")[0]


# Kernels that do the same job for different layer shapes are one ROLE: the launcher picks among them per layer, so the
# share of a step that "conv3x3 forward + input gradient" takes is the sum over all three of its kernels.
ROLES = (
    ("conv3x3 fwd+dgrad (conv3_wide + conv3_ws64 + conv3_patch_tn)", ("conv3_wide_kernel", "conv3_ws64_kernel", "conv3_patch_tn_kernel")),
    ("conv3x3 wgrad (conv3_wgrad_patch)", ("conv3_wgrad_patch_kernel", "conv3_wgrad_pc_kernel", "conv3_wgrad_multi_kernel")),
    ("batchnorm bwd (bn_act_bwd + bn_bwd_conv1)", ("bn_act_bwd(reduce+apply)", "bn_bwd_conv1_kernel")),
    ("batchnorm fwd (bn_act_fwd)", ("bn_act_fwd_kernel",)),
)


def role(name):
    f = family(name)
    for r, members in ROLES:
        if f in members:
            return r
    return f


def by_family(kernels):
    """fold the per-kernel rows of profile_pass into one row per role, largest share of the step first."""
    fam = {}
    for r in kernels:
        f = fam.setdefault(role(r["kernel"]), {"kernel": role(r["kernel"]), "bound": r["bound"], "ms_per_step": 0.0,
                                               "launches_per_step": 0.0, "work": 0.0, "peak": r["peak"], "unit": r["unit"]})
        f["ms_per_step"] += r["ms_per_step"]
        f["launches_per_step"] += r["launches_per_step"]
        f["work"] += r["achieved"] * r["ms_per_step"]          # (flop or byte rate) x time = work per step
    out = []
    for f in fam.values():
        f["achieved"] = f["work"] / f["ms_per_step"] if f["ms_per_step"] > 0 else 0.0
        f["frac"] = f["achieved"] / f["peak"]
        f["avg_us"] = 1e3 * f["ms_per_step"] / max(f["launches_per_step"], 1e-9)
        del f["work"]
        out.append(f)
    out.sort(key=lambda r: -r["ms_per_step"])
    return out


def roofline_of(kernels, traffic_config=""):
    """the `roofline` object: the role with the largest share of a step in this instrumented pass."""
    top = by_family(kernels)[0]
    return {"kernel": top["kernel"], "bound": top["bound"], "achieved": top["achieved"], "peak": top["peak"],
            "unit": top["unit"], "frac": top["frac"], "traffic": pmc_traffic(top["kernel"], traffic_config),
            "avg_us": top["avg_us"], "launches_per_step": top["launches_per_step"]}


def profile_pass(step_fn, batches, nsteps=3):
    """instrumented pass: per-kernel HIP-event timings aggregated by the library."""
    from vae_amd import _lib
    _lib.call("vae_prof_start", 1 << 14)
    for i in range(nsteps):
        step_fn(batches[i % len(batches)])
    rows = 128
    names = ctypes.create_string_buffer(64 * rows)
    launches = (ctypes.c_int * rows)()
    ms = (ctypes.c_double * rows)()
    fl = (ctypes.c_double * rows)()
    by = (ctypes.c_double * rows)()
    n = _lib.call("vae_prof_stop", names, launches, ms, fl, by, rows)
    out = []
    for k in range(min(n, rows)):
        name = names.raw[64 * k:64 * (k + 1)].split(b"\0")[0].decode()
        t = ms[k] * 1e-3
        mfma = name.startswith(("gemm_", "conv3_"))      # the matrix-core kernels; everything else streams HBM
        rec = {"kernel": name, "launches_per_step": launches[k] / nsteps, "ms_per_step": ms[k] / nsteps,
               "avg_us": 1e3 * ms[k] / max(launches[k], 1), "bound": "mfma" if mfma else "hbm"}
        if mfma:
            peak = PEAK_TFLOPS_FP32 if name.startswith("gemm_f32") else PEAK_TFLOPS_BF16
            rec.update(achieved=fl[k] / t / 1e12 if t > 0 else 0.0, peak=peak, unit="TFLOP/s")
        else:
            rec.update(achieved=by[k] / t / 1e9 if t > 0 else 0.0, peak=PEAK_HBM_GBS, unit="GB/s")
        rec["frac"] = rec["achieved"] / rec["peak"]
        rec["gbs"] = by[k] / t / 1e9 if t > 0 else 0.0
        out.append(rec)
    out.sort(key=lambda r: -r["ms_per_step"])
    return out


def pmc_traffic(kernel, config):
    """HBM bytes per launch of the role `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE are
    collected in separate profiler runs of this script and folded by tools_pmc_traffic.py; they cannot be
    read from inside the timed process).  None when no measurement of this workload's kernels is on file."""
    import glob
    name = {"cfg2_1gpu": "r*_pmc_traffic.json", "cfg4_celeba64": "r*_pmc_traffic_cfg4.json"}.get(config)
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", name))) if name else []
    if not paths:
        return None
    try:
        with open(paths[-1]) as f:                 # the latest round's measurement
            ks = json.load(f)["kernels"]
        # (the two-pass rows cover two kernel launches per call: tools_pmc_traffic.py already reports bytes per call)
        rows = [(v, v.get("launches_profiled", 1) / (2 if k.startswith(("bn_act_bwd", "head_bwd")) else 1))
                for k, v in ks.items() if role(k) == kernel]
        n = sum(w for _, w in rows)
        return sum(v["hbm_bytes_per_launch"] * w for v, w in rows) / n if n else None
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(budget_s=15.0):
    """the CPU oracle's full train step at BASELINE config 1 (batch 64, fp32), timed on this host."""
    from oracle import vae_oracle as O
    torch.manual_seed(1337)
    sd = O.construct_state(3, 3, 128, 0.1, 32)
    opt = O.new_opt_state(sd)
    g = torch.Generator().manual_seed(1337)
    x = torch.rand(64, 3, 32, 32, generator=g)
    shapes = [(64, 64, 32, 32), (64, 128, 16, 16), (64, 256, 8, 8), (64, 256, 4, 4)]

    def one():
        eps = torch.randn(64, 128, generator=g)
        masks = [(torch.rand(s, generator=g) >= 0.1).to(torch.uint8) for s in shapes]
        O.train_step(x, sd, opt, eps, "l1", 1.0, 0.1, masks)

    one(); one()                                   # warm-up
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < budget_s or n < 5:
        one()
        n += 1
    dt = time.perf_counter() - t0
    return {"value": 64 * n / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"config 1 (VAEConv(3,3,128,0.1,32), batch 64, fp32, full train step) x {n} steps in {dt:.1f} s; "
                      f"host has {os.cpu_count()} logical CPUs",
            "ms_per_step": 1e3 * dt / n}


def build_model(config, precision, rank, dev, side_stream=False):
    """(conf, model) of a benchmark config: random-init VAEConv of that architecture on `dev`, train mode."""
    from vae_amd.train import load_config
    from vae_amd.vae import VAEConv
    conf = load_config(config, os.path.join(ROOT, "config"))
    S, Lat, Cin = int(conf.image_size), int(conf.latent_dims), int(conf.num_channels)
    torch.manual_seed(int(conf.seed))
    model = VAEConv(Cin, 3, Lat, float(conf.p_dropout), S).to(dev).train()
    model.rng_seed, model.rank = int(conf.seed), rank
    model.use_side_stream = bool(side_stream)
    model.set_precision(precision)
    return conf, model


def timed_loop(fn, items, steps, warmup, world, dev, per_step_events=False):
    """`warmup` untimed calls, then exactly `steps` calls of fn(item) between barrier + synchronize on both sides;
    returns (seconds: max over ranks, last result, per-step event times in ms or None)."""
    import torch.distributed as dist
    out = None
    for i in range(warmup):
        out = fn(items[i % len(items)])
    # one event after every step (on torch's current stream, the one every kernel of the step is launched on):
    # the per-step durations behind the median; recording costs ~1 us of host time per step
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if per_step_events else None
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if marks:
        marks[0].record()
    for i in range(steps):
        out = fn(items[i % len(items)])
        if marks:
            marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps)) if marks else None
    return elapsed, out, per_step


def run_decode(config, batch, steps, warmup, world, rank, dev, profile=True):
    """sampling throughput: recon = model.decode(z), eval mode, z ~ N(0,1) resident in HBM; replicas only
    (no collective): every rank decodes its own batches."""
    conf, model = build_model(config, "bf16", rank, dev)
    from vae_amd.dp import broadcast_state
    broadcast_state(model)
    model.eval()
    S, Lat = int(conf.image_size), int(conf.latent_dims)
    batch = batch or int(conf.get("decode_batch", 0)) or int(conf.batch_size)
    g = torch.Generator(device=dev).manual_seed(int(conf.seed) + rank)
    zs = [torch.randn(batch, Lat, device=dev, generator=g) for _ in range(8)]
    with torch.no_grad():
        elapsed, out, _ = timed_loop(model.decode, zs, steps, warmup, world, dev)
        kernels = profile_pass(model.decode, zs) if profile else None
    rec = {"metric": "z->image decodes/sec (whole node)", "value": world * batch * steps / elapsed, "unit": "images/s",
           "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"{config}: VAEConv decode(z), latent {Lat}, {S}x{S}, eval mode, batch {batch} per GPU",
                      "global_batch": world * batch, "parallelism": f"replicas x{world}"},
           "checksum": float(out.float().mean())}
    if kernels:
        rec["roofline"] = roofline_of(kernels)
        rec["kernel_families"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in by_family(kernels)[:8]]
    return rec


def run_train(config, batch, precision, steps, warmup, world, rank, dev, side_stream=False, torch_optimizer=False,
              profile=True, traffic=True):
    """full train steps of a benchmark config on synthetic batches resident in HBM; the contract's JSON fields."""
    import torch.distributed as dist
    from vae_amd.dp import BucketedReducer, broadcast_state, stage_ranges
    from vae_amd.optim import FlatAdamW
    from vae_amd.train import FusedStep
    conf, model = build_model(config, precision, rank, dev, side_stream)
    broadcast_state(model)
    batch_cfg = int(conf.batch_size)
    batch = batch or batch_cfg
    S, Lat, Cin = int(conf.image_size), int(conf.latent_dims), int(conf.num_channels)
    opt = torch.optim.AdamW(model.parameters(), lr=float(conf.lr)) if torch_optimizer else FlatAdamW(model, lr=float(conf.lr))
    reducer = None
    if world > 1:
        reducer = BucketedReducer(model.flat_grads, stage_ranges(model._offsets, model.flat_grads.numel()))
    step = FusedStep(model, opt, conf, reducer)
    g = torch.Generator(device=dev).manual_seed(int(conf.seed) + rank)
    batches = [torch.rand(batch, Cin, S, S, device=dev, generator=g) for _ in range(8)]
    n_ranks_seen = dist.get_world_size() if world > 1 else 1
    elapsed, loss, per_step = timed_loop(step, batches, steps, warmup, world, dev, per_step_events=True)
    final_loss = loss.item()
    kernels = None
    if profile:
        plain = FusedStep(model, opt, conf, None)        # no collectives in the instrumented pass
        kernels = profile_pass(plain, batches)
    ips = world * batch * steps / elapsed
    fpi = train_flops_per_image(Cin, Lat, S)
    peak = PEAK_TFLOPS_BF16 if precision == "bf16" else PEAK_TFLOPS_FP32
    out = {
        "metric": "training images/sec (whole node)", "value": ips, "unit": "images/s", "n_gpus": world,
        "n_ranks_seen": n_ranks_seen,
        "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
        "ms_per_step_median": per_step[len(per_step) // 2], "ms_per_step_min": per_step[0],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if precision == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": f"{config}: VAEConv({Cin},3,{Lat},{float(conf.p_dropout)},{S}) full train step "
                               f"(fwd + {conf.recon_loss}/KL loss + bwd + clip + AdamW), batch {batch} per GPU",
                   "global_batch": world * batch, "parallelism": f"dp{world}"},
        "model_tflops": ips * fpi / 1e12,
        "model_mfma_frac": ips * fpi / 1e12 / (peak * world),
        "final_loss": final_loss,
    }
    if kernels:
        fams = by_family(kernels)
        out["kernel_families"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in fams[:12]]
        out["roofline"] = roofline_of(kernels, config if (traffic and batch == batch_cfg) else "")
        out["kernels"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()} for r in kernels[:24]]
        out["kernel_ms_per_step_sum"] = sum(r["ms_per_step"] for r in kernels)
    return out


# the other single-GPU workloads of the benchmark configs: (label, kind, config, precision, steps, warmup)
SECONDARY = (
    ("cfg4 train", "train", "cfg4_celeba64", "bf16", 12, 4),
    ("cfg5 train", "train", "cfg5_beta", "bf16", 30, 8),
    ("cfg5 decode", "decode", "cfg5_beta", "bf16", 30, 8),
    ("cfg1 fp32 mode", "train", "cfg1_cpu", "fp32", 12, 4),
)


def run_secondary(dev):
    """each of SECONDARY as a short run of its own (a few seconds together); a failure is recorded, not raised: the
    headline line must still be printed."""
    out = []
    for label, kind, config, precision, steps, warmup in SECONDARY:
        try:
            if kind == "decode":
                r = run_decode(config, 0, steps, warmup, 1, 0, dev)
            else:
                r = run_train(config, 0, precision, steps, warmup, 1, 0, dev)
            keep = ("metric", "value", "unit", "dtype", "ms_per_step", "steps", "warmup", "config", "model_tflops",
                    "model_mfma_frac", "roofline")
            out.append({"label": label, **{k: r[k] for k in keep if k in r}})
        except Exception as e:          # noqa: BLE001 -- report and go on
            out.append({"label": label, "config": {"workload": config}, "error": f"{type(e).__name__}: {e}"})
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` typed as is: this process (which has made no GPU call and makes none) starts one
    rank per GPU under torch.distributed.run as a child process and hands its exit status on."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def parse_tune(items):
    """--tune KEY=INT -> [(key, value)]; anything else is a usage error (the library range-checks the value)."""
    out = []
    for kv in items:
        k, sep, v = kv.partition("=")
        try:
            out.append((k.strip(), int(v)))
        except ValueError:
            sep = ""
        if not sep or not k.strip():
            raise SystemExit(f"--tune expects KEY=INT (e.g. --tune conv3_wide=1), got {kv!r}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["train", "decode"], default="train")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="cfg2_1gpu")
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU batch size")
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="bf16",
                    help="bf16: the matrix-core path (the headline).  fp32: the library's fp32 kernels, the reference's own "
                         "arithmetic type (e.g. --config cfg1_cpu --precision fp32 = the reference's CPU configuration on the GPU)")
    ap.add_argument("--side-stream", action="store_true",
                    help="run the weight-gradient kernels of the backward pass on a second stream")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=INT",
                    help="vae_set_tuning(KEY, INT) before the run (measurement aid), e.g. --tune conv3_wide=1; the keys and "
                         "their ranges are listed at vae_set_tuning in include/vae_hip.h")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo to rehearse ranks that share a GPU)")
    ap.add_argument("--watchdog", type=int, default=0, help="dump every thread's stack and exit after this many seconds (0 = off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-profile", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short runs of the other single-GPU configs")
    ap.add_argument("--torch-optimizer", action="store_true", help="torch.optim.AdamW + clip_grad_norm_ instead of the fused kernel")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="form the process group, count the ranks with one all-reduce, print that and leave (touches no GPU "
                         "with --backend gloo: checks the launcher path on a machine without one)")
    args = ap.parse_args()
    tune = parse_tune(args.tune)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    if args.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=True)

    import torch.distributed as dist
    from vae_amd import _lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE): they must agree")
    if args.rendezvous_only:
        if world > 1:
            if args.backend == "nccl":
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(args.backend)
        one = torch.ones(1, device=torch.device("cuda", local_rank) if args.backend == "nccl" else "cpu")
        if world > 1:
            dist.all_reduce(one)
        if rank == 0:
            print(json.dumps({"rendezvous": "ok", "n_gpus": args.gpus, "n_ranks_seen": int(one.item()), "backend": args.backend}))
        if world > 1:
            dist.destroy_process_group()
        return
    if args.backend != "nccl":
        local_rank %= torch.cuda.device_count()      # a rehearsal: several ranks may share one card
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
    _lib.load()
    for k, v in tune:
        _lib.call("vae_set_tuning", k.encode(), v)

    profile = not args.no_kernel_profile
    if args.mode == "decode":
        out = run_decode(args.config, args.batch, args.steps, args.warmup, world, rank, dev, profile)
    else:
        out = run_train(args.config, args.batch, args.precision, args.steps, args.warmup, world, rank, dev,
                        args.side_stream, args.torch_optimizer, profile)
        default_run = (world == 1 and args.config == "cfg2_1gpu" and not args.batch and args.precision == "bf16"
                       and not tune and not args.torch_optimizer)
        if default_run and profile and not args.no_secondary:
            torch.cuda.empty_cache()
            out["secondary"] = run_secondary(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
