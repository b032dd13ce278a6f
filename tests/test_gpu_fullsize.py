"""The benchmark configurations at their FULL sizes (batch 512 at 32x32, latent 128; batch 1024 at 64x64, latent 256).

At batch 512 / 32x32 the CPU oracle still finishes a forward pass in seconds, so the train-mode forward, the loss and the gradients are
compared with it directly (eps and dropout masks from the Philox streams, regenerated on the host).  At batch 1024 / 64x64 the
oracle's forward pass ran on the build machine and its digests are a committed fixture (tests/golden/b64_full1024.npz); beyond that
the checks there are properties that do not depend on the size: the pass is bit-reproducible,
an eval-mode batch equals the concatenation of its halves to fp32 rounding (no kernel may mix samples outside
BatchNorm's batch statistics), staged and whole backward agree, and a training step moves the loss down."""
import pytest
import torch

from oracle import vae_oracle as O
from golden_util import rel_l2, zero_grad_bias

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _model(latent, size, seed=1337):
    from vae_amd.vae import VAEConv
    torch.manual_seed(seed)
    m = VAEConv(3, 3, latent, 0.1, size).to(DEV)
    m.rng_seed, m.rank = seed, 0
    return m


@pytest.mark.parametrize("L,kind,klw", [(128, "l1", 1.0), (512, "mse", 4.0)], ids=["cfg2", "cfg5"])
def test_cfg2_cfg5_forward_and_loss_against_the_oracle(L, kind, klw):
    """benchmark configs 2 (latent 128, L1 + KL) and 5 (latent 512, beta = 4, MSE + KL) at their full batch of 512"""
    N, S, p = 512, 32, 0.1
    m = _model(L, S).train()
    m.rng_step = 5
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = torch.rand(N, 3, S, S, generator=torch.Generator().manual_seed(7))
    recon, mu, logvar = m(x.to(DEV))
    eps = torch.from_numpy(O.philox_normal(N * L, 1337, O.STREAM_EPS, 5, 0)).view(N, L)
    masks = []
    for b, (C, H) in enumerate([(64, S), (128, S // 2), (256, S // 4), (256, S // 8)]):
        k = O.philox_keep_mask(N * H * H * C, p, 1337, b, 5, 0)
        masks.append(torch.from_numpy(k).view(N, H, H, C).permute(0, 3, 1, 2))
    loss_ref, r_ref, mu_ref, lv_ref, g_ref, _ = O.loss_and_grads(x, sd, eps, kind, klw, True, p, masks)
    # the contract's bf16 figure on the outputs; mu / logvar sit behind eight bf16 conv + BatchNorm blocks
    assert rel_l2(recon.detach().cpu(), r_ref) <= 2e-2
    assert rel_l2(mu.detach().cpu(), mu_ref) <= 5e-2
    assert rel_l2(logvar.detach().cpu(), lv_ref) <= 5e-2
    from vae_amd.train import Config, vae_loss
    loss = vae_loss(recon, x.to(DEV), mu, logvar, kind, None, Config(kl_weight=klw, percep_weight=0.0))
    assert abs(float(loss.detach()) - float(loss_ref)) <= 1e-2 * abs(float(loss_ref))
    # gradients: bf16 rounding flips a few ReLU / max-pool routes per thousand, which no bf16 pipeline reproduces element
    # by element; over a batch of 512 the flips average out and direction and size of every tensor's gradient must agree
    loss.backward()
    worst, dot, n_got, n_ref = 1.0, 0.0, 0.0, 0.0
    for name, prm in m.named_parameters():
        ref = g_ref[name].flatten().double()
        got = prm.grad.detach().cpu().flatten().double()
        if zero_grad_bias(name):          # a bias in front of a train-mode BatchNorm: analytically zero, rounding noise in the oracle
            assert float(got.abs().max()) <= 1e-4 and float(ref.abs().max()) <= 1e-4, name
            continue
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm()))
        ratio = float(got.norm() / ref.norm())
        worst = min(worst, cos)
        dot, n_got, n_ref = dot + float(torch.dot(got, ref)), n_got + float(got.norm() ** 2), n_ref + float(ref.norm() ** 2)
        # the first block, eight bf16 blocks deep, is the worst.  Two bf16-faithful evaluations of this function that differ
        # only in fp32 summation order are 0.13-0.16 apart in relative L2 on these tensors (test_oracle_bf16_noise_floor.py),
        # so a norm ratio is only pinned to about +-15 %: swapping one conv kernel for another moved the 64-entry
        # conv1.net.4.bias from 1.09 to 1.106.
        assert cos >= 0.90 and 0.85 <= ratio <= 1.15, (name, cos, ratio)
    total = dot / (n_got * n_ref) ** 0.5
    print(f"latent {L}: gradient cosine at batch 512: all parameters", total, "worst tensor", worst)
    assert total >= 0.995


def test_cfg4_full_size_forward_and_loss_against_the_oracle_fixture():
    """Benchmark config 4 at its full size (batch 1024, 64x64, latent 256, dropout 0.1, train mode): the fp32 oracle
    needs minutes and ~20 GB for it, so it ran on the build machine (oracle/make_fullsize_fixture.py) and left
    digests -- norm, sum and 16384 strided samples of recon / mu / logvar, the loss, the per-image L1 sums.  Weights,
    input, eps and the dropout masks are regenerated here from the same seeds (the module's own Philox streams)."""
    import os
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "b64_full1024.npz"))
    N, L, S, seed, step, xseed = (int(v) for v in z["cfg"])
    m = _model(L, S, seed).train()
    m.rng_step = step
    x = torch.rand(N, 3, S, S, generator=torch.Generator().manual_seed(xseed))
    xd = x.to(DEV)
    recon, mu, logvar = m(xd)
    from vae_amd.train import Config, l1_kl_percep_loss
    loss = l1_kl_percep_loss(recon, xd, mu, logvar, Config(kl_weight=1.0, percep_weight=0.0), None)
    ref_loss = float(z["loss/l1"])
    assert abs(float(loss.detach()) - ref_loss) <= 1e-2 * abs(ref_loss), (float(loss.detach()), ref_loss)
    for name, t, tol in (("recon", recon, 2e-2), ("mu", mu, 5e-2), ("logvar", logvar, 5e-2)):
        dig, smp = O.tensor_digest(t.detach().float().cpu(), 16384)
        ref_dig, ref_smp = z[f"{name}/digest"], z[f"{name}/sample"]
        assert abs(dig[0] - ref_dig[0]) <= tol * ref_dig[0], (name, dig[0], ref_dig[0])
        err = np.linalg.norm(smp.astype(np.float64) - ref_smp) / np.linalg.norm(ref_smp.astype(np.float64))
        print(f"cfg4 full size: {name} relative L2 error on {ref_smp.size} samples {err:.4f}, norm {dig[0]:.3f} vs {ref_dig[0]:.3f}")
        assert err <= tol, (name, err)
    per_image = (recon.detach() - xd).abs().sum(dim=(1, 2, 3)).cpu().numpy()
    assert np.abs(per_image - z["per_image_l1"]).max() <= 2e-2 * np.abs(z["per_image_l1"]).max()


@pytest.mark.parametrize("mode,tol", [("bf16", 2e-2), ("fp32", 1e-5)])
def test_cfg5_decode_eval_mode_against_the_oracle(mode, tol):
    """Benchmark config 5's sampling leg at its full size: recon = decode(z) in eval mode (vae/models/vae.py:114-122;
    running statistics, no dropout, no RNG), latent 512, batch 4096, 32x32, against the fp32 oracle's decoder -- 2e-2 on
    the bf16 path, 1e-5 in the fp32 mode.  Affine BatchNorm parameters and running statistics are moved off their
    initial (1, 0, 0, 1) so that they matter."""
    from vae_amd.vae import VAEConv
    N, L, S = 4096, 512, 32
    torch.manual_seed(1337)
    m = VAEConv(3, 3, L, 0.1, S)
    sd = O.perturb_norm_params({k: v.detach().clone() for k, v in m.state_dict().items()}, 5)
    m.load_state_dict(sd)
    m.to(DEV).eval().set_precision(mode)
    z = torch.randn(N, L, generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        ref = O.decode(z, sd, train=False)
        got = m.decode(z.to(DEV)).cpu()
    assert got.shape == (N, 3, S, S) and torch.isfinite(got).all()
    err = rel_l2(got, ref)
    worst = float((got - ref).abs().max())
    print(f"cfg5 decode, batch {N}, {mode}: relative L2 {err:.2e}, largest absolute difference {worst:.2e}")
    assert err <= tol, (mode, err)
    assert worst <= (3e-2 if mode == "bf16" else 2e-6)
    # nothing moved: eval mode leaves the running statistics and the counters alone
    after = m.state_dict()
    for k in sd:
        if "running" in k or "num_batches" in k:
            assert torch.equal(after[k].cpu(), sd[k]), k


@pytest.mark.parametrize("N,L,S", [(512, 128, 32), (1024, 256, 64), (512, 512, 32)])
def test_full_size_properties(N, L, S):
    from vae_amd.optim import FlatAdamW
    from vae_amd.train import Config, FusedStep
    m = _model(L, S)
    x = torch.rand(N, 3, S, S, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3))

    # eval mode: samples are independent, so a batch is the concatenation of its halves
    m.eval()
    with torch.no_grad():
        mu, lv = m.encode(x)
        mu_a, lv_a = m.encode(x[: N // 2])
        mu_b, lv_b = m.encode(x[N // 2:])
        # (the latent heads split K by the batch size, so their fp32 sums are reordered: equal to rounding, not to the bit)
        assert rel_l2(torch.cat([mu_a, mu_b]).cpu(), mu.cpu()) <= 1e-5
        assert rel_l2(torch.cat([lv_a, lv_b]).cpu(), lv.cpu()) <= 1e-5
        z = torch.randn(N, L, device=DEV, generator=torch.Generator(device=DEV).manual_seed(4))
        r = m.decode(z)
        assert rel_l2(torch.cat([m.decode(z[: N // 2]), m.decode(z[N // 2:])]).cpu(), r.cpu()) <= 1e-5
        assert r.shape == (N, 3, S, S) and float(r.min()) > 0 and float(r.max()) < 1

    # train mode: the same step number gives the same bits, whole or staged backward
    m.train()
    conf = Config(kl_weight=1.0, percep_weight=0.0, recon_loss="l1", lr=1e-4)
    opt = FlatAdamW(m, lr=1e-4)
    fs = FusedStep(m, opt, conf, None)
    from vae_amd import _lib

    def grads(staged):
        m.rng_step = 11
        recon, mu_, lv_, state = m.forward_raw(x, None, None)
        d_recon, partial, out3, grid = fs._buffers(recon, x.device)
        _lib.call("vae_recon_loss_fwd_bwd", recon.data_ptr(), x.data_ptr(), d_recon.data_ptr(), partial.data_ptr(),
                  recon.numel(), fs.kind, 1.0 / N, _lib.current_stream())
        m.flat_grads.zero_()
        for stage in ((0, 1, 2) if staged else (-1,)):
            m.backward_raw(state, d_recon, None, None, 1.0 / N, stage)
        m._release_workspace(state["batch"], state["ws"])
        torch.cuda.synchronize()
        return m.flat_grads.clone(), recon.clone()

    g1, r1 = grads(False)
    g2, r2 = grads(False)
    g3, r3 = grads(True)
    assert torch.equal(r1, r2) and torch.equal(g1, g2), "the pass is not reproducible"
    assert torch.equal(g1, g3), "staged and whole backward differ"
    assert torch.isfinite(g1).all() and float(g1.abs().sum()) > 0

    # fused steps on one batch move the loss down (eps and dropout are redrawn every step, so compare ends, not neighbours)
    losses = [float(fs(x)) for _ in range(30)]
    assert all(torch.isfinite(torch.tensor(losses))) and max(losses[-5:]) < losses[0]


@pytest.mark.parametrize("N,S", [(1, 32), (2, 32), (3, 32), (5, 32), (7, 32), (12, 32), (33, 32), (100, 32), (1, 64), (3, 64), (17, 64)])
def test_launch_forms_agree_at_any_batch(N, S):
    """the same pass with the conv3x3 weight gradients of a stage in one launch or one launch per layer, and with the
    first conv in two passes or stored: same loss inputs, gradients equal to the order of their sums (1e-4 per tensor);
    covers the K-split planner and the slab carving at batches from 1 up (every workgroup budget from 'one K tile' on)"""
    from vae_amd import _lib
    from vae_amd.optim import FlatAdamW
    from vae_amd.train import Config, FusedStep
    m = _model(64, S)
    m.train()
    x = torch.rand(N, 3, S, S, device=DEV, generator=torch.Generator(device=DEV).manual_seed(N + S))
    fs = FusedStep(m, FlatAdamW(m, lr=1e-4), Config(kl_weight=1.0, percep_weight=0.0, recon_loss="l1", lr=1e-4), None)

    def grads(staged=False):
        m.rng_step = 5
        recon, mu_, lv_, state = m.forward_raw(x, None, None)
        d_recon, partial, out3, grid = fs._buffers(recon, x.device)
        _lib.call("vae_recon_loss_fwd_bwd", recon.data_ptr(), x.data_ptr(), d_recon.data_ptr(), partial.data_ptr(),
                  recon.numel(), fs.kind, 1.0 / N, _lib.current_stream())
        m.flat_grads.zero_()
        for stage in ((0, 1, 2) if staged else (-1,)):
            m.backward_raw(state, d_recon, None, None, 1.0 / N, stage)
        m._release_workspace(state["batch"], state["ws"])
        torch.cuda.synchronize()
        return [g.clone() for g in m.grad_views()], recon.clone()

    base, r0 = grads()
    staged, _ = grads(staged=True)
    assert all(torch.equal(a, b) for a, b in zip(base, staged))
    names = [k for k, _ in m.named_parameters()]
    for key in (b"wgrad_batch", b"conv1_twice"):
        _lib.call("vae_set_tuning", key, 1)
        try:
            other, r1 = grads()
        finally:
            _lib.call("vae_set_tuning", key, 0)
        assert torch.equal(r0, r1), key                               # the forward output does not depend on either
        for k, a, b in zip(names, base, other):
            assert torch.isfinite(a).all() and torch.isfinite(b).all(), k
            assert rel_l2(a.cpu(), b.cpu()) <= 1e-4, (key, k, rel_l2(a.cpu(), b.cpu()))


@pytest.mark.parametrize("N,L,S,p", [(6, 64, 24, 0.1), (3, 64, 40, 0.0), (130, 64, 16, 0.2), (1, 128, 32, 0.1), (2, 128, 128, 0.1)])
def test_ragged_shapes_against_the_oracle(N, L, S, p):
    """image sizes that are not powers of two (the generic kernels), odd batches, a batch of one: train-mode forward,
    loss and gradients against the oracle with the Philox streams regenerated on the host"""
    from vae_amd.vae import VAEConv
    from vae_amd.train import Config, l1_kl_percep_loss
    torch.manual_seed(5)
    m = VAEConv(3, 3, L, p, S).to(DEV).train()
    m.rng_seed, m.rank, m.rng_step = 99, 2, 7
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = torch.rand(N, 3, S, S, generator=torch.Generator().manual_seed(N + S))
    recon, mu, logvar = m(x.to(DEV))
    eps = torch.from_numpy(O.philox_normal(N * L, 99, O.STREAM_EPS, 7, 2)).view(N, L)
    masks = None
    if p > 0:
        masks = []
        for b, (C, H) in enumerate([(64, S), (128, S // 2), (256, S // 4), (256, S // 8)]):
            k = O.philox_keep_mask(N * H * H * C, p, 99, b, 7, 2)
            masks.append(torch.from_numpy(k).view(N, H, H, C).permute(0, 3, 1, 2))
    loss_ref, r_ref, mu_ref, lv_ref, g_ref, _ = O.loss_and_grads(x, sd, eps, "l1", 1.0, True, p, masks)
    assert rel_l2(recon.detach().cpu(), r_ref) <= 2e-2
    if N > 1:       # a batch of one at 4x4 leaves BatchNorm 16 values per channel: mu / logvar are noise-dominated
        assert rel_l2(mu.detach().cpu(), mu_ref) <= 8e-2
    loss = l1_kl_percep_loss(recon, x.to(DEV), mu, logvar, Config(kl_weight=1.0, percep_weight=0.0), None)
    assert abs(float(loss.detach()) - float(loss_ref)) <= 2e-2 * abs(float(loss_ref))
    loss.backward()
    dot = n_got = n_ref = 0.0
    for name, prm in m.named_parameters():
        if zero_grad_bias(name):
            continue
        got, ref = prm.grad.detach().cpu().flatten().double(), g_ref[name].flatten().double()
        assert torch.isfinite(got).all(), name
        dot, n_got, n_ref = dot + float(torch.dot(got, ref)), n_got + float(got.norm() ** 2), n_ref + float(ref.norm() ** 2)
    total = dot / (n_got * n_ref) ** 0.5
    print(f"N={N} S={S}: gradient cosine over all parameters {total:.4f}, norm ratio {(n_got / n_ref) ** 0.5:.3f}")
    assert total >= (0.97 if N > 1 else 0.85) and 0.85 <= (n_got / n_ref) ** 0.5 <= 1.15


def test_single_channel_input_against_the_oracle():
    """num_channels: 1 (the MNIST option of the reference's config/local.yaml:12-13): forward against the oracle"""
    from vae_amd.vae import VAEConv
    torch.manual_seed(9)
    m = VAEConv(1, 3, 64, 0.0, 32).to(DEV).train()
    m.rng_seed, m.rank, m.rng_step = 5, 0, 0
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = torch.rand(16, 1, 32, 32, generator=torch.Generator().manual_seed(2))
    recon, mu, logvar = m(x.to(DEV))
    eps = torch.from_numpy(O.philox_normal(16 * 64, 5, O.STREAM_EPS, 0, 0)).view(16, 64)
    with torch.no_grad():
        r_ref, mu_ref, lv_ref = O.forward(x, sd, eps, True, 0.0, None)
    assert recon.shape == (16, 3, 32, 32)
    assert rel_l2(recon.detach().cpu(), r_ref) <= 2e-2 and rel_l2(mu.detach().cpu(), mu_ref) <= 5e-2


@pytest.mark.parametrize("N,L,S,Cin,Cout", [(5, 64, 16, 3, 3), (3, 64, 32, 4, 3), (3, 64, 32, 3, 1), (2, 64, 48, 3, 3), (2, 192, 32, 3, 3)])
def test_other_sizes_and_channel_counts_against_the_oracle(N, L, S, Cin, Cout):
    """2x2 bottleneck (16x16 images), four input channels, one output channel, a 6x6 bottleneck, a latent size that is
    not a power of two: train-mode forward against the oracle"""
    from vae_amd.vae import VAEConv
    torch.manual_seed(1)
    m = VAEConv(Cin, Cout, L, 0.0, S).to(DEV).train()
    m.rng_seed, m.rank, m.rng_step = 3, 0, 0
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    x = torch.rand(N, Cin, S, S, generator=torch.Generator().manual_seed(S))
    recon, mu, logvar = m(x.to(DEV))
    eps = torch.from_numpy(O.philox_normal(N * L, 3, O.STREAM_EPS, 0, 0)).view(N, L)
    with torch.no_grad():
        r_ref, mu_ref, lv_ref = O.forward(x, sd, eps, True, 0.0, None)
    assert m.chw == (256, S // 8, S // 8) and recon.shape == (N, Cout, S, S)
    assert rel_l2(recon.detach().cpu(), r_ref) <= 2e-2 and rel_l2(mu.detach().cpu(), mu_ref) <= 5e-2
    assert rel_l2(logvar.detach().cpu(), lv_ref) <= 8e-2


def test_image_size_8_fails_like_the_reference():
    """a 1x1 bottleneck: the constructor's train-mode dummy forward (vae/models/vae.py:76-86) runs BatchNorm on one
    value per channel, which torch refuses; the same error is raised here"""
    from vae_amd.vae import VAEConv
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        VAEConv(3, 3, 64, 0.0, 8)
