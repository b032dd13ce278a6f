"""world_size-2 CPU (gloo) tests of the data-parallel exchange: per-stage bucketed all-reduce
of the flat gradient buffer gives every rank the mean gradient, bucket by bucket, and the
gradient exchange of a data-parallel step (each rank: loss and grads of its own half batch with
per-rank BatchNorm statistics, then the mean over ranks in backward-stage order) leaves both
ranks with the same buffer, equal to the single-process mean of the per-rank gradients.  The
oracle supplies the per-rank arithmetic so the test needs no GPU."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import vae_oracle as O
from vae_amd import dp

WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _flat_layout(shapes):
    names = [k for k in shapes if not k.endswith(("running_mean", "running_var", "num_batches_tracked"))]
    offsets, o = [], 0
    for k in names:
        n = 1
        for d in shapes[k]:
            n *= d
        offsets.append((o, n))
        o += n
    return names, offsets, o


def _rank_grads(rank, state, shapes):
    """loss and grads of this rank's half of a fixed global batch (eval-free, dropout off)."""
    torch.manual_seed(1234)
    x = torch.rand(8, 3, 16, 16)
    eps = torch.randn(8, 8)
    xs, es = x[rank * 4:(rank + 1) * 4], eps[rank * 4:(rank + 1) * 4]
    loss, _, _, _, grads, _ = O.loss_and_grads(xs, state, es, kind="l1", kl_weight=1.0, train=True, p_dropout=0.0)
    return loss, grads


def _worker(rank, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        torch.set_num_threads(2)
        # ---- 1. plain reducer semantics on a synthetic buffer
        total = 1000
        # one exchange per stage (default), and the first stage's range riding with the second's
        for merge, counts in ((False, [4, 3, 2]), (True, [0, 6, 2])):
            g = torch.arange(total, dtype=torch.float32) * (rank + 1)
            red = dp.BucketedReducer(g, [(600, 1000), (250, 600), (0, 250)], bucket_bytes=4 * 128, merge_first=merge)
            assert [len(b) for b in red.buckets] == counts
            for s in range(3):
                red.stage_done(s)
                if merge and s == 0:
                    assert red.handles == []
            red.wait()
            expect = torch.arange(total, dtype=torch.float32) * 1.5
            assert torch.equal(g, expect)
            assert red.handles == []

        # ---- 2. a data-parallel step with the oracle's arithmetic
        shapes = O.state_shapes(3, 3, 8, 16)
        names, offsets, n_total = _flat_layout(shapes)
        torch.manual_seed(99)
        state = O.construct_state(3, 3, 8, 0.0, 16)
        # same start on every rank (what broadcast_state does for the GPU model)
        for k in sorted(state):
            dist.broadcast(state[k], 0)
        loss, grads = _rank_grads(rank, state, shapes)
        flat = torch.zeros(n_total)
        for k, (o, n) in zip(names, offsets):
            flat[o:o + n] = grads[k].reshape(-1)
        red = dp.BucketedReducer(flat, dp.stage_ranges(offsets, n_total), bucket_bytes=64 << 10)
        for s in range(3):              # backward order: decoder+heads, conv4+conv3, conv2+conv1
            red.stage_done(s)
        red.wait()
        # ---- 3. state replication helpers on a stand-in for the module's flat buffers
        class Stub:
            def __init__(self):
                self.flat_params = torch.full((10,), float(rank + 1))
                self._bn_stats = torch.arange(6, dtype=torch.float32) * (rank + 1)
                self._nbt = torch.tensor([3 + rank])
                self.repacked = 0

            def repack_weights(self):
                self.repacked += 1
        m = Stub()
        dp.broadcast_state(m)
        assert torch.equal(m.flat_params, torch.full((10,), 1.0)) and int(m._nbt) == 3 and m.repacked == 1
        assert torch.equal(m._bn_stats, torch.arange(6, dtype=torch.float32))
        m._bn_stats = torch.arange(6, dtype=torch.float32) * (rank + 1)      # running statistics drift apart per rank
        dp.sync_bn_stats(m)
        assert torch.equal(m._bn_stats, torch.arange(6, dtype=torch.float32) * 1.5)
        # ---- 4. fit() on two ranks: the running statistics are averaged BEFORE the validation pass, so every rank
        # validates the same model and the checkpoint rank 0 writes is the model whose loss was compared.  The GPU
        # steps are replaced by stand-ins (this process has no GPU); the epoch loop itself is the code under test.
        import vae_amd.data as vdata
        import vae_amd.train as vtrain
        seen = {"val_stats": [], "saved": []}

        class Toy(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.w = torch.nn.Parameter(torch.zeros(1))
                self._bn_stats = torch.zeros(4)

        toy = Toy()

        def fake_train_step(model, optimizer, x, config):
            model._bn_stats += float(rank + 1)               # per-rank batch statistics: the running stats drift apart
            return torch.tensor(1.0)

        def fake_val_step(model, x, config):
            seen["val_stats"].append(model._bn_stats.clone())
            return model._bn_stats.sum() * 0 + 2.0 - 0.1 * len(seen["val_stats"])      # improves every epoch

        def fake_save(model, path):
            seen["saved"].append(model._bn_stats.clone())

        old = vtrain.train_step, vtrain.val_step, vdata.save_checkpoint
        vtrain.train_step, vtrain.val_step, vdata.save_checkpoint = fake_train_step, fake_val_step, fake_save
        try:
            conf = vtrain.Config(n_epochs=2, save_model=True, percep_weight=0.0, kl_weight=1.0)
            hist = vtrain.fit(toy, None, conf, [torch.zeros(1)] * 3, [torch.zeros(1)], fused=False)
        finally:
            vtrain.train_step, vtrain.val_step, vdata.save_checkpoint = old
        # epoch 1: ranks added 3*1 and 3*2 -> mean 4.5; epoch 2: 4.5 + 3*(rank+1) -> mean 9.0
        assert [float(t[0]) for t in seen["val_stats"]] == [4.5, 9.0], seen["val_stats"]
        assert len(hist) == 2 and hist[1]["val/loss"] < hist[0]["val/loss"]
        assert ([float(t[0]) for t in seen["saved"]] == [4.5, 9.0]) if rank == 0 else (seen["saved"] == [])
        torch.save({"flat": flat, "loss": loss.detach()}, os.path.join(outdir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_reducer_two_ranks_gloo():
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(port, d), nprocs=WORLD, join=True)
        out = [torch.load(os.path.join(d, f"r{r}.pt"), weights_only=True) for r in range(WORLD)]
    # every rank holds the same reduced gradient
    assert torch.equal(out[0]["flat"], out[1]["flat"])
    # and it is the mean of the per-rank gradients, computed here in one process
    shapes = O.state_shapes(3, 3, 8, 16)
    names, offsets, n_total = _flat_layout(shapes)
    torch.manual_seed(99)
    state = O.construct_state(3, 3, 8, 0.0, 16)
    ref = torch.zeros(n_total)
    losses = []
    for r in range(WORLD):
        loss, grads = _rank_grads(r, state, shapes)
        losses.append(loss.detach())
        for k, (o, n) in zip(names, offsets):
            ref[o:o + n] += grads[k].reshape(-1) / WORLD
    # (fp32 summation order differs with the thread count of the workers, hence a norm-relative bar)
    assert float((out[0]["flat"] - ref).norm() / ref.norm()) <= 1e-5
    for r in range(WORLD):
        assert torch.allclose(out[r]["loss"], losses[r], rtol=1e-5)
    assert float(ref.abs().sum()) > 0
