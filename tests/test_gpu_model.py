"""End-to-end parity of the MI355X VAEConv (vae_amd.vae / vae_amd.train) against the golden
vectors produced by the reference classes and against the CPU oracle.

The GPU path computes the GEMM-shaped layers in bf16 with fp32 accumulation, so the bar is
the bf16 figure of the hot-path contract: 2e-2 relative (per-tensor L2 error / L2 norm) on
outputs, loss and gradients for identical inputs, weights, eps and dropout masks."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vae_oracle as O
from golden_util import Case, digest_rel_err, rel_l2, zero_grad_bias

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF16_RTOL = 2e-2
LOSS_FNS = {"l1": F.l1_loss, "mse": F.mse_loss, "bce": F.binary_cross_entropy,
            "bce_logits": "bce_logits"}     # the draft's clamped form, as the fixture was made (oracle/make_fixtures.py)


def build(case):
    from vae_amd.vae import VAEConv
    torch.manual_seed(case.seed)
    m = VAEConv(**case.cfg)
    return m


def nhwc_masks(masks):
    return None if masks is None else [m.permute(0, 2, 3, 1).contiguous().to(DEV) for m in masks]


class Cfg(dict):
    __getattr__ = dict.__getitem__


@pytest.mark.parametrize("name", ["a32_train", "a32_dropout", "b64_train", "c32_beta4"])
def test_construction_matches_reference(name):
    c = Case(name)
    m = build(c)
    sd = m.state_dict()
    ref = c.init_state()
    assert list(sd.keys()) == list(ref.keys()) and len(sd) == 85
    for k in ref:
        assert sd[k].shape == ref[k].shape, k
        assert torch.equal(sd[k].float(), ref[k].float()), k     # same seed -> same init as the oracle restatement
    assert m.chw == (256, c.cfg["image_size"] // 8, c.cfg["image_size"] // 8)
    assert m.flattened_dims == 256 * (c.cfg["image_size"] // 8) ** 2
    # moving to the GPU keeps names, shapes and values; parameters become views of one buffer
    m.to(DEV)
    sd2 = m.state_dict()
    assert list(sd2.keys()) == list(ref.keys())
    for k in ref:
        assert torch.equal(sd2[k].cpu().float(), ref[k].float()), k
    assert m.flat_params.numel() >= sum(p.numel() for p in m.parameters())


# What an end-to-end comparison of a bf16 pipeline can and cannot show (measured):
#  * Every kernel is checked on its own, tightly, in test_gpu_kernels*.py (same bf16 operands ->
#    only fp32 summation order and one output rounding differ).
#  * End to end, bf16 rounding noise is AMPLIFIED by the network itself: each conv+BatchNorm block
#    roughly doubles the relative error (mean subtraction), and a relative activation error e flips
#    a fraction ~e of the ReLU / max-pool routing decisions, which moves a gradient tensor by
#    ~sqrt(2e).  Two bf16 runs that differ only in fp32 summation order therefore drift apart to
#    the bf16 noise level within a few layers; oracle.forward_bf16 (the CPU restatement with the
#    kernels' own rounding points) differs from the fp32 reference by 3 % at mu/logvar and by
#    10-35 % on encoder gradients at these random weights, and so does the HIP path.
#  * So: loss and reconstruction are held to the 2e-2 bf16 figure against the fp32 reference;
#    mu/logvar to the measured bf16 noise level; shallow gradients (head, bn1) to 2e-2; every
#    gradient tensor must point the same way as both the bf16 oracle's and the fp32 reference's
#    (cosine) with the same magnitude -- a wiring, layout or scaling bug gives cosine ~0 or a wrong norm.
# Measured (profiles/r02_parity.json, tools_parity_report.py) and asserted at about 1.5x the measurement, as RELATIVE L2:
#   vs the bf16-faithful oracle: recon 1.0-1.2e-3, mu / logvar 0.3-1.0e-2, worst gradient tensor 0.11-0.20, all
#   gradients together 0.06-0.09 on the 2-4 image fixtures (1.5e-2 at batch 512);
#   vs the fp32 reference:       recon 1.3-1.5e-3, mu / logvar 2.6-3.7e-2, worst gradient tensor 0.33-0.40.
# The same bf16 oracle evaluated with a renumbered channel order (identical operands and mathematics, another fp32
# summation order) differs from ITSELF by recon 1.1e-3, mu / logvar 0.7-1.0e-2, worst gradient tensor 0.13-0.16
# (tests/test_oracle_bf16_noise_floor.py): that is the floor of any comparison between two bf16 pipelines.
EMU_RECON_RTOL = 2e-3
EMU_DEEP_RTOL = 1.6e-2
DEEP_RTOL = 5e-2
GRAD_REL_EMU = 0.30          # per gradient tensor, vs the bf16 oracle
GRAD_REL_EMU_ALL = 0.13      # all gradients as one vector, vs the bf16 oracle
GRAD_REL_REF = 0.60          # per gradient tensor, vs the fp32 reference
GRAD_COS_EMU = 0.97
# against the fp32 reference the bar sits below the worst case of the bf16 oracle itself: on the 4-image fixtures
# a 64-entry BatchNorm bias gradient of the first block lands at 0.90 +- 0.01 depending on the summation order
GRAD_COS_REF = 0.85


def cosine(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-30)).item()


@pytest.mark.parametrize("name", ["a32_train", "a32_dropout", "a32_eval", "b64_train", "c32_beta4"])
def test_forward_loss_backward_vs_golden(name):
    from vae_amd.train import vae_loss
    c = Case(name)
    m = build(c)
    m.load_state_dict(c.state())
    m.to(DEV)
    m.train(c.train)
    x, eps = c.inputs()
    xd, epsd = x.to(DEV), eps.to(DEV)
    cpu_masks = c.masks()
    masks = nhwc_masks(cpu_masks)
    p = c.cfg["p_dropout"]
    sd_before = {k: v.clone() for k, v in m.state_dict().items()}
    first = True
    for kind in c.kinds():
        m.load_state_dict(sd_before)
        m.zero_grad()
        if c.train:
            recon, mu, logvar = m(xd, epsd, masks)
            e_loss, e_recon, e_mu, e_lv, e_grads, e_stats = O.loss_and_grads_bf16(
                x, c.state(), eps, kind, c.kl_weight, p, cpu_masks)
        else:
            with torch.no_grad():
                recon, mu, logvar = m(xd, epsd)
                e_recon, e_mu, e_lv = O.forward_bf16(x, c.state(), eps, False, p, None)
                e_loss = O.vae_loss(e_recon, x, e_mu, e_lv, kind, c.kl_weight)
        conf = Cfg(percep_weight=0.0, kl_weight=c.kl_weight)
        loss = vae_loss(recon, xd, mu, logvar, LOSS_FNS[kind], None, conf)
        ref_loss = c.z[f"loss/{kind}"].item()
        assert abs(loss.item() - ref_loss) <= BF16_RTOL * abs(ref_loss), (kind, loss.item(), ref_loss)
        assert abs(loss.item() - e_loss.item()) <= EMU_RECON_RTOL * abs(e_loss.item()), (kind, loss.item(), e_loss.item())
        if first:
            # fp32 reference
            assert rel_l2(recon.cpu(), c.z["recon"]) <= BF16_RTOL
            assert rel_l2(mu.cpu(), c.z["mu"]) <= DEEP_RTOL
            assert rel_l2(logvar.cpu(), c.z["logvar"]) <= DEEP_RTOL
            # bf16-faithful oracle
            assert rel_l2(recon.cpu(), e_recon) <= EMU_RECON_RTOL
            assert rel_l2(mu.cpu(), e_mu) <= EMU_DEEP_RTOL
            assert rel_l2(logvar.cpu(), e_lv) <= EMU_DEEP_RTOL
            if c.train:
                after = m.state_dict()
                for k in c.names("stats_after"):
                    if k.endswith("num_batches_tracked"):
                        assert int(after[k].item()) == int(sd_before[k].item()) + 1
                        continue
                    r, _, _ = digest_rel_err(c, "stats_after", k, after[k], 64)
                    assert r <= BF16_RTOL, (k, r)
                    assert rel_l2(after[k].cpu(), e_stats[k]) <= EMU_DEEP_RTOL, k
        if c.train:
            loss.backward()
            ref_grads = O.loss_and_grads(x, c.state(), eps, kind, c.kl_weight, True, p, cpu_masks)[4]
            report = []
            for k, pp in m.named_parameters():
                assert pp.grad is not None, k
                g = pp.grad.detach().cpu()
                r, abs_rms, ref_rms = digest_rel_err(c, f"grad/{kind}", k, pp.grad, 1024 if first else 128)
                if zero_grad_bias(k):
                    assert abs_rms <= 1e-4, (k, abs_rms)       # analytically zero; reference holds ~1e-6 noise
                    assert g.abs().max().item() == 0.0
                    continue
                ce, cr = cosine(g, e_grads[k]), cosine(g, ref_grads[k])
                ne, nr = g.norm().item() / e_grads[k].norm().item(), g.norm().item() / ref_grads[k].norm().item()
                report.append((k, rel_l2(g, e_grads[k]), ce, ne, r, cr, nr))
                assert ce >= GRAD_COS_EMU and abs(ne - 1) <= 0.1, (k, ce, ne)
                assert cr >= GRAD_COS_REF and abs(nr - 1) <= 0.15, (k, cr, nr)
                assert rel_l2(g, e_grads[k]) <= GRAD_REL_EMU, (k, rel_l2(g, e_grads[k]))
                assert rel_l2(g, ref_grads[k]) <= GRAD_REL_REF, (k, rel_l2(g, ref_grads[k]))
            flat = lambda d: torch.cat([d[k].reshape(-1).double() for k, _, *_ in report])
            got_all = torch.cat([pp.grad.detach().cpu().reshape(-1).double() for k, pp in m.named_parameters() if not zero_grad_bias(k)])
            assert rel_l2(got_all, flat(e_grads)) <= GRAD_REL_EMU_ALL, rel_l2(got_all, flat(e_grads))
            wk = max(report, key=lambda t: t[1])
            print(f"{name}/{kind}: worst gradient vs bf16 oracle: {wk[0]} rel {wk[1]:.3f} cos {wk[2]:.4f}; "
                  f"vs fp32 reference rel {wk[4]:.3f} cos {wk[5]:.4f}")
            by_name = {t[0]: t[4] for t in report}
            for k in ("head.weight", "head.bias", "bn1.weight", "bn1.bias"):
                assert by_name[k] <= BF16_RTOL, (k, by_name[k])
        first = False


def test_train_steps_vs_golden_and_fused_equals_autograd():
    from vae_amd.train import FusedStep, train_step
    c = Case("a32_steps")
    steps = int(c.z["cfg/steps"].item())
    conf = Cfg(percep_weight=0.0, kl_weight=1.0, recon_loss="l1")
    results = []
    for mode in ("autograd", "fused"):
        m = build(c)
        m.load_state_dict(c.state())
        m.to(DEV).train()
        opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
        step = FusedStep(m, opt, conf) if mode == "fused" else None
        losses = []
        for s in range(steps):
            x, eps = c.inputs(c.seed + 100 * s)
            xd, epsd = x.to(DEV), eps.to(DEV)
            loss = step(xd, epsd) if step else train_step(m, opt, xd, conf, epsd)
            losses.append(loss.item())
        results.append((losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}))
        for s in range(steps):
            ref = c.z[f"step{s}/loss"].item()
            assert abs(losses[s] - ref) <= BF16_RTOL * abs(ref), (mode, s, losses[s], ref)
        for k, v in results[-1][1].items():
            if k.endswith("num_batches_tracked"):
                # the constructor's dummy forward already counted once for the encoder BatchNorms
                assert int(v.item()) == steps + (1 if k.startswith("conv") else 0), k
                continue
            r, abs_rms, ref_rms = digest_rel_err(c, "final", k, v.float())
            # parameters move by ~lr per step; compare the value (loose) -- the strict check is the loss trajectory
            # AdamW's first steps move every weight by ~lr in the direction of sign(grad), so the
            # bf16 gradient noise shows up in the weights and, through them, in the deep running stats
            tol = 0.15 if "running_" in k else BF16_RTOL
            assert r <= tol or abs_rms <= 3.5e-4, (mode, k, r, abs_rms)
    (l0, s0), (l1, s1) = results
    assert np.allclose(l0, l1, rtol=1e-6), (l0, l1)
    for k in s0:
        assert torch.allclose(s0[k].float(), s1[k].float(), rtol=1e-5, atol=1e-7), k


def test_philox_driven_forward_matches_oracle():
    """No injection: eps and the dropout masks come from the module's Philox streams; the same
    streams regenerated on the host feed the oracle."""
    c = Case("a32_dropout")
    m = build(c)
    m.load_state_dict(c.state())
    m.to(DEV).train()
    m.rng_seed, m.rng_step, m.rank = 4242, 17, 3
    x, _ = c.inputs()
    N, L, S, p = c.n, c.cfg["latent_dims"], c.cfg["image_size"], c.cfg["p_dropout"]
    recon, mu, logvar = m(x.to(DEV))
    eps = torch.from_numpy(O.philox_normal(N * L, 4242, O.STREAM_EPS, 17, 3)).view(N, L)
    masks = []
    for b, (C, H) in enumerate([(64, S), (128, S // 2), (256, S // 4), (256, S // 8)]):
        k = O.philox_keep_mask(N * H * H * C, p, 4242, b, 17, 3)
        masks.append(torch.from_numpy(k).view(N, H, H, C).permute(0, 3, 1, 2))
    with torch.no_grad():
        r_ref, mu_ref, lv_ref = O.forward(x, c.state(), eps, True, p, masks)
    assert rel_l2(recon.detach().cpu(), r_ref) <= BF16_RTOL
    assert rel_l2(mu.detach().cpu(), mu_ref) <= DEEP_RTOL
    assert rel_l2(logvar.detach().cpu(), lv_ref) <= DEEP_RTOL
    assert m.rng_step == 18


def test_encode_decode_and_errors():
    c = Case("a32_eval")
    m = build(c)
    m.load_state_dict(c.state())
    x, eps = c.inputs()
    with pytest.raises(RuntimeError, match="MI355X"):
        m(x)                                    # module and input on the CPU: no fallback
    m.to(DEV).eval()
    with torch.no_grad():
        mu, logvar = m.encode(x.to(DEV))
        z = mu + torch.exp(0.5 * logvar) * eps.to(DEV)
        recon = m.decode(z)
    assert rel_l2(mu.cpu(), c.z["mu"]) <= BF16_RTOL
    assert rel_l2(recon.cpu(), c.z["recon"]) <= BF16_RTOL
    with pytest.raises(RuntimeError, match="expected input"):
        m(torch.zeros(2, 3, 16, 16, device=DEV))
    z2 = m.reparameterize(mu, logvar)
    assert z2.shape == mu.shape


def test_packed_weights_follow_torch_writes_to_parameters():
    """The kernels read bf16 packed copies of the weights.  Any torch-side write to a parameter -- a torch
    optimizer's step (the reference's own AdamW, vae/training/train.py:310,338), load_state_dict after a forward,
    p.copy_() -- must be seen by the next pass: its output has to equal a freshly built module holding the same
    state_dict, bit for bit."""
    from vae_amd.train import FusedStep, train_step
    c = Case("a32_train")
    conf = Cfg(percep_weight=0.0, kl_weight=1.0, recon_loss="l1")
    x, eps = c.inputs()
    xd, epsd = x.to(DEV), eps.to(DEV)

    def fresh_output(sd):
        f = build(c)
        f.load_state_dict(sd)
        f.to(DEV).eval()
        with torch.no_grad():
            return [t.clone() for t in f(xd, epsd)]

    def own_output(m):
        m.eval()
        with torch.no_grad():
            out = [t.clone() for t in m(xd, epsd)]
        m.train()
        return out

    for mode in ("autograd", "fused-torch-optimizer"):
        m = build(c)
        m.load_state_dict(c.state())
        m.to(DEV).train()
        opt = torch.optim.AdamW(m.parameters(), lr=2e-2)          # large steps: stale weights cannot hide in bf16 noise
        step = FusedStep(m, opt, conf) if mode != "autograd" else None
        before = own_output(m)
        for _ in range(3):
            step(xd, epsd) if step else train_step(m, opt, xd, conf, epsd)
        sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        got, want = own_output(m), fresh_output(sd)
        assert rel_l2(got[0].cpu(), before[0].cpu()) > 1e-2, mode     # the weights did move
        for a, b in zip(got, want):
            assert torch.equal(a, b), mode
    # load_state_dict on a module that has already run (and packed)
    m.load_state_dict(c.state())
    for a, b in zip(own_output(m), fresh_output(c.state())):
        assert torch.equal(a, b)
    # an in-place write to one parameter
    with torch.no_grad():
        m.conv2.net[3].weight.mul_(1.5)
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for a, b in zip(own_output(m), fresh_output(sd)):
        assert torch.equal(a, b)


def test_alternative_kernel_paths_give_the_same_gradients():
    """The whole-model backward has two forms of three passes (`vae_set_tuning`): the first conv's BatchNorm backward +
    weight gradient in one pass or as separate kernels, the pooled BatchNorm sums from the stored window maxima or from
    the windows, the head's sums on the matrix pipe or in vector registers.  Same inputs, same Philox streams: the
    window-maxima form is bit-identical, the other two differ only by where sums are rounded (the one-pass first conv
    does not round the gradient to bf16 before summing: its weight gradient moves by the bf16 rounding of the other;
    the head's sums move in the last bit, and what follows them down the network by the bf16 noise floor)."""
    from vae_amd import _lib
    c = Case("a32_dropout")
    x, _ = c.inputs()

    def grads(**keys):
        for k, v in keys.items():
            _lib.call("vae_set_tuning", k.encode(), v)
        try:
            m = build(c)
            m.load_state_dict(c.state())
            m.to(DEV).train()
            m.rng_seed, m.rng_step, m.rank = 99, 3, 0
            recon, mu, logvar = m(x.to(DEV))
            loss = F.l1_loss(recon, x.to(DEV), reduction="sum") / c.n + (-0.5 * (1 + logvar - mu ** 2 - logvar.exp()).sum()) / c.n
            loss.backward()
            return {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
        finally:
            for k in keys:
                _lib.call("vae_set_tuning", k.encode(), 0)

    base = grads()
    again = grads()
    assert all(torch.equal(base[n], again[n]) for n in base)
    ymax_off = grads(pool_ymax=1)
    assert all(torch.equal(base[n], ymax_off[n]) for n in base)
    head_alt = grads(head_sums=1)
    conv1_alt = grads(conv1_bwd=1)
    first = ("conv1.net.0.weight", "conv1.net.0.bias", "conv1.net.1.weight", "conv1.net.1.bias")
    names = list(base)
    head_own = [n for n in names if n.startswith(("head.", "bn1."))]
    assert len(head_own) == 4
    for n in names:
        # the first conv is the end of the backward chain: nothing else may move
        if n not in first:
            assert torch.equal(conv1_alt[n], base[n]), n
        elif not zero_grad_bias(n):
            assert rel_l2(conv1_alt[n], base[n]) <= 2e-2, n
        # the head is its start: its own sums agree to fp32 rounding; a last-bit change of the coefficients then flips
        # bf16 roundings of the gradient it hands down, which the layers below amplify to the bf16 noise floor
        if zero_grad_bias(n):
            continue
        assert rel_l2(head_alt[n], base[n]) <= (2e-5 if n in head_own else 0.2), n
    # the last ConvTranspose's backward: the head's gradient formed inside it (default), stored as bf16 and read by the
    # one-kernel backward (2), or by the two GEMMs (1).  The head's own sums do not depend on it; the two stored forms
    # differ by fp32 summation order in the weight gradient only; the default form rounds the head's gradient to bf16
    # from a differently associated expression (measured: 5e-6 on up1.weight, 1.2e-2 at the far end of the encoder)
    stored, gemms = grads(convt_bwd=2), grads(convt_bwd=1)
    for n in names:
        if zero_grad_bias(n):
            continue
        assert rel_l2(gemms[n], stored[n]) <= 1e-5, n
        assert rel_l2(stored[n], base[n]) <= (2e-5 if n in head_own else 1e-3 if n == "up1.weight" else 5e-2), n
