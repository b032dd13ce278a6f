"""GPU side of the data-parallel step (one process; the exchange itself is covered on CPU by
tests/test_dp_gloo.py): a gradient range handed to the reducer after backward stage s must already
hold its final values, the three staged calls must give the same gradients as the single call, and
FusedStep must call the reducer once per stage, in order, before the optimizer runs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(batch=8, size=16):
    from vae_amd.optim import FlatAdamW
    from vae_amd.train import Config
    from vae_amd.vae import VAEConv
    torch.manual_seed(1337)
    model = VAEConv(3, 3, 64, 0.1, size).to("cuda").train()
    model.rng_seed, model.rank = 1337, 0
    conf = Config(kl_weight=1.0, percep_weight=0.0, recon_loss="l1", lr=1e-3)
    opt = FlatAdamW(model, lr=1e-3)
    g = torch.Generator(device="cuda").manual_seed(77)
    x = torch.rand(batch, 3, size, size, device="cuda", generator=g)
    return model, conf, opt, x


def _loss_grad(model, x, fs):
    from vae_amd import _lib
    recon, mu, logvar, state = model.forward_raw(x, None, None)
    d_recon, partial, out3, grid = fs._buffers(recon, x.device)
    _lib.call("vae_recon_loss_fwd_bwd", recon.data_ptr(), x.data_ptr(), d_recon.data_ptr(), partial.data_ptr(),
              recon.numel(), fs.kind, 1.0 / x.size(0), _lib.current_stream())
    return state, d_recon


def test_staged_backward_ranges_are_final_when_handed_over():
    from vae_amd.dp import stage_ranges
    from vae_amd.train import FusedStep
    model, conf, opt, x = _make()
    fs = FusedStep(model, opt, conf, None)
    ranges = stage_ranges(model._offsets, model.flat_grads.numel())

    state, d_recon = _loss_grad(model, x, fs)
    model.backward_raw(state, d_recon, None, None, 1.0 / x.size(0), -1)
    whole = model.flat_grads.clone()
    assert float(whole.abs().sum()) > 0

    # same forward state, staged: poison the buffer's unfinished ranges to catch late writes into finished ones
    snaps = []
    for stage in (0, 1, 2):
        model.backward_raw(state, d_recon, None, None, 1.0 / x.size(0), stage)
        a, b = ranges[stage]
        snaps.append(model.flat_grads[a:b].clone())
    model._release_workspace(state["batch"], state["ws"])
    torch.cuda.synchronize()
    assert torch.equal(model.flat_grads, whole)
    for stage, (a, b) in enumerate(ranges):
        assert torch.equal(snaps[stage], whole[a:b]), f"stage {stage} range changed after it was handed over"
    # the three ranges tile the buffer
    assert sorted(ranges)[0][0] == 0 and sorted(ranges)[-1][1] == whole.numel()


def test_side_stream_backward_gives_the_same_gradients():
    """weight-gradient kernels on a second stream (event-ordered, three rotating gradient buffers): bit-identical
    to the single-stream pass, whole and staged, three times over (a race would show as a difference)"""
    from vae_amd.train import FusedStep
    model, conf, opt, x = _make(batch=256, size=32)     # large enough for the two streams to really overlap
    fs = FusedStep(model, opt, conf, None)
    state, d_recon = _loss_grad(model, x, fs)
    assert state["cfg"].side_stream is None
    model.backward_raw(state, d_recon, None, None, 1.0 / x.size(0), -1)
    torch.cuda.synchronize()
    ref = model.flat_grads.clone()
    side = torch.cuda.Stream()
    state["cfg"].side_stream = side.cuda_stream
    for rep in range(3):
        model.flat_grads.zero_()
        if rep == 1:
            for stage in (0, 1, 2):
                model.backward_raw(state, d_recon, None, None, 1.0 / x.size(0), stage)
        else:
            model.backward_raw(state, d_recon, None, None, 1.0 / x.size(0), -1)
        torch.cuda.synchronize()
        assert torch.equal(model.flat_grads, ref), f"repetition {rep}"
    model._release_workspace(state["batch"], state["ws"])
    # and through the module switch
    model.use_side_stream = True
    assert model._cfg(8, True).side_stream is not None and model._cfg(8, False).side_stream is None


class _RecordingReducer:
    """stands in for BucketedReducer: records the call order and the gradient range at hand-over"""

    def __init__(self, flat, ranges):
        self.flat, self.ranges, self.calls, self.snaps = flat, ranges, [], {}

    def stage_done(self, stage):
        a, b = self.ranges[stage]
        self.calls.append(("stage", stage))
        self.snaps[stage] = self.flat[a:b].clone()

    def wait(self):
        self.calls.append(("wait",))
        self.final = self.flat.clone()


def test_fused_step_drives_the_reducer_in_stage_order():
    from vae_amd.dp import stage_ranges
    from vae_amd.train import FusedStep
    model, conf, opt, x = _make()
    ranges = stage_ranges(model._offsets, model.flat_grads.numel())
    red = _RecordingReducer(model.flat_grads, ranges)
    before = model.flat_params.clone()
    loss = FusedStep(model, opt, conf, red)(x)
    torch.cuda.synchronize()
    assert red.calls == [("stage", 0), ("stage", 1), ("stage", 2), ("wait",)]
    for stage, (a, b) in enumerate(ranges):
        assert torch.equal(red.snaps[stage], red.final[a:b])
    assert float(loss) > 0 and not torch.equal(model.flat_params, before)

    # identical to the step without a reducer (world size 1 semantics)
    model2, conf2, opt2, x2 = _make()
    loss2 = FusedStep(model2, opt2, conf2, None)(x2)
    torch.cuda.synchronize()
    assert float(loss2) == float(loss)
    assert torch.equal(model2.flat_params, model.flat_params)
