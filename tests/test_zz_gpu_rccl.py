"""The data-parallel step over RCCL itself, as far as one GPU allows: a process group of one rank with the
"nccl" backend, the staged backward handing its gradient ranges to the bucketed reducer on its side stream, the
exchange forced on.  The mean over one rank is the identity, so losses and parameters must equal the run without
a reducer bit for bit -- which checks the stream hand-over (main -> exchange stream -> RCCL's stream -> main), the
AVG reduction op and the in-place bucket views, everything but the second rank.  (Collected last: it owns the
default process group while it runs.)"""
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_reducer_over_rccl_is_the_identity_at_world_size_one():
    from vae_amd.dp import BucketedReducer, stage_ranges
    from vae_amd.optim import FlatAdamW
    from vae_amd.train import Config, FusedStep
    from vae_amd.vae import VAEConv
    if dist.is_initialized():
        pytest.skip("a default process group already exists in this process")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        conf = Config(kl_weight=1.0, percep_weight=0.0, recon_loss="l1", lr=1e-4)
        xs = [torch.rand(128, 3, 32, 32, device="cuda", generator=torch.Generator(device="cuda").manual_seed(i)) for i in range(3)]

        def run(with_reducer, side_stream=False, merge_first=False):
            torch.manual_seed(1337)
            m = VAEConv(3, 3, 128, 0.1, 32).cuda().train()
            m.rng_seed, m.rank = 1337, 0
            red = None
            if with_reducer:
                red = BucketedReducer(m.flat_grads, stage_ranges(m._offsets, m.flat_grads.numel()), bucket_bytes=2 << 20,
                                      side_stream=side_stream, merge_first=merge_first)
                assert red.avg_op and sum(len(b) for b in red.buckets) >= 6 and bool(red.buckets[0]) != merge_first
                red.world = 2          # force the exchange although there is nobody to exchange with
            fs = FusedStep(m, FlatAdamW(m, lr=1e-4), conf, red)
            losses = torch.stack([fs(xs[i % 3]).clone() for i in range(12)])
            torch.cuda.synchronize()
            return losses.cpu(), m.flat_params.clone().cpu()

        la, pa = run(False)
        assert torch.isfinite(la).all()
        # issued from the compute stream (default) or from a stream of its own; first stage exchanged alone or with the second
        for side_stream, merge_first in ((False, False), (True, False), (False, True)):
            lb, pb = run(True, side_stream, merge_first)
            assert torch.equal(la, lb) and torch.equal(pa, pb), (side_stream, merge_first)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_step_stays_on_the_single_process_loss_curve():
    """300 steps of the bf16 run of tests/test_gpu_elbo_match.py through the bucketed reducer of a one-rank RCCL group
    (exchange forced on): the same losses and the same validation loss, bit for bit -- the data-parallel step is on
    the curve that test compares with the fp32 mode."""
    from tools_elbo_match import run_curve
    from vae_amd.dp import BucketedReducer, stage_ranges
    if dist.is_initialized():
        pytest.skip("a default process group already exists in this process")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        def reducer(m):
            red = BucketedReducer(m.flat_grads, stage_ranges(m._offsets, m.flat_grads.numel()))
            red.world = 2          # force the exchange although there is nobody to exchange with
            return red
        la, va = run_curve("bf16", "l1", 300, 128)
        lb, vb = run_curve("bf16", "l1", 300, 128, reducer_factory=reducer)
        assert torch.isfinite(la).all() and float(la[-20:].mean()) < float(la[:20].mean())
        assert torch.equal(la, lb) and va == vb
    finally:
        dist.destroy_process_group()
