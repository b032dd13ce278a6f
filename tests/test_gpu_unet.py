"""The drafts' U-Net VAE (vae/practice/vae_original.py:77-181) over the kernels: the 1x1 logits head and the
ConvTranspose node against torch, the whole model against the fixtures made from the reference class and against
the oracle's gradients."""
import pytest
import torch
import torch.nn.functional as F

from oracle import vae_oracle as O
from tests.golden_util import rel_l2
from tests.test_oracle_unet import UnetCase

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def L():
    from vae_amd import _lib
    _lib.load()
    return _lib


def bf(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("n,H,W,Co", [(2, 4, 4, 3), (3, 5, 7, 1), (1, 32, 32, 4), (70, 16, 16, 3)])
def test_conv1x1_out_fwd_bwd(L, n, H, W, Co):
    g = torch.Generator().manual_seed(n + H + Co)
    x = bf(torch.randn(n, 64, H, W, generator=g)).requires_grad_(True)
    w = (torch.randn(Co, 64, 1, 1, generator=g) / 8).requires_grad_(True)
    b = torch.randn(Co, generator=g).requires_grad_(True)
    ref = F.conv2d(x, w, b)
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    wd, bd = w.detach().reshape(Co, 64).contiguous().to(DEV), b.detach().to(DEV)
    out = torch.full((n, Co, H, W), float("nan"), device=DEV)
    L.call("vae_conv1x1_out_fwd", L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(out), n, H, W, Co, L.current_stream())
    assert rel_l2(out.cpu(), ref.detach()) <= 1e-5
    nb = L.call("vae_conv1x1_out_bwd_blocks", n, H, W)
    partial = torch.full((nb, (Co + 1) * 64), float("nan"), device=DEV)
    sums = torch.full(((Co + 1) * 64,), float("nan"), device=DEV)
    dx = torch.full((n, H, W, 64), float("nan"), device=DEV, dtype=torch.bfloat16)
    gd = gout.to(DEV)
    L.call("vae_conv1x1_out_bwd", L.ptr(gd), L.ptr(xd), L.ptr(wd), L.ptr(partial), L.ptr(sums), L.ptr(dx), n, H, W, Co,
           L.current_stream())
    assert rel_l2(dx.float().cpu().permute(0, 3, 1, 2), x.grad) <= 4e-3          # bf16 store
    assert rel_l2(sums[:Co * 64].cpu().view(Co, 64, 1, 1), w.grad) <= 1e-5
    assert rel_l2(sums[Co * 64:Co * 64 + Co].cpu(), b.grad) <= 1e-5
    with pytest.raises(RuntimeError, match="Cout=2"):
        L.call("vae_conv1x1_out_fwd", L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(out), n, H, W, 2, L.current_stream())


@pytest.mark.parametrize("n,H,W,Ci,Co", [(2, 4, 4, 256, 128), (3, 8, 8, 128, 64), (1, 16, 16, 64, 64)])
def test_upconv_node_against_autograd(n, H, W, Ci, Co):
    from vae_amd.unet import UpConvT2x2
    torch.manual_seed(n + Ci)
    up = UpConvT2x2(Ci, Co)
    with torch.no_grad():
        up.weight.copy_(bf(up.weight))
    x = bf(torch.randn(n, Ci, H, W)).requires_grad_(True)
    ref = F.conv_transpose2d(x, up.weight, up.bias, stride=2)
    gout = bf(torch.randn(ref.shape))
    gx, gw, gb = torch.autograd.grad(ref, (x, up.weight, up.bias), gout)
    up = up.to(DEV)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV).requires_grad_(True)
    y = up.forward_nhwc(xd)
    assert rel_l2(y.float().cpu().permute(0, 3, 1, 2), ref.detach()) <= 4e-3
    y.backward(gout.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV))
    assert rel_l2(xd.grad.float().cpu().permute(0, 3, 1, 2), gx) <= 4e-3
    assert rel_l2(up.weight.grad.cpu(), gw) <= 1e-4
    assert rel_l2(up.bias.grad.cpu(), gb) <= 1e-5


def test_clamp_f32(L):
    g = torch.Generator().manual_seed(2)
    x = (4 * torch.randn(1000, generator=g)).requires_grad_(True)
    x.data[:4] = torch.tensor([-5.0, 5.0, -5.0001, 5.0001])           # the bounds themselves pass the gradient
    ref = torch.clamp(x, min=-5, max=5)
    go = torch.randn(1000, generator=g)
    ref.backward(go)
    xd, gd = x.detach().to(DEV), go.to(DEV)
    out = torch.full((1000,), float("nan"), device=DEV)
    L.call("vae_clamp_f32", L.ptr(xd), None, -5.0, 5.0, L.ptr(out), 1000, L.current_stream())
    assert torch.equal(out.cpu(), ref.detach())
    L.call("vae_clamp_f32", L.ptr(xd), L.ptr(gd), -5.0, 5.0, L.ptr(out), 1000, L.current_stream())
    assert torch.equal(out.cpu(), x.grad)


def cosine(a, b):
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-300))


@pytest.mark.parametrize("name", ["unet256_s16", "unet256_s32_nodrop", "unet1024_s32"])
def test_unet_against_reference_fixture(name):
    from vae_amd import unet
    from vae_amd.unet import vae_loss_mse
    c = UnetCase(name)
    torch.manual_seed(c.seed)
    m = getattr(unet, "VAEUnet" + c.variant)(3, c.latent, c.p, c.size)
    torch.manual_seed(c.seed)
    init = O.construct_state_unet(c.variant, c.latent, c.p, c.size)
    msd = m.state_dict()
    assert list(msd.keys()) == list(init.keys())
    assert all(torch.equal(msd[k], init[k]) for k in init)          # same seed, same initial weights as the reference
    sd = c.state()
    m.load_state_dict(sd)
    m = m.to(DEV).train()
    x, eps = c.inputs()
    masks = c.masks()
    out, mu, logvar = m(x.to(DEV), eps=eps.to(DEV), keep_masks={k: v.to(DEV) for k, v in masks.items()})
    # bf16 kernels against the fp32 reference: 2e-2 relative
    assert rel_l2(out.detach().cpu(), c.z["out"]) <= 2e-2
    assert rel_l2(mu.detach().cpu(), c.z["mu"]) <= 2e-2
    assert rel_l2(logvar.detach().cpu(), c.z["logvar"]) <= 2e-2
    loss = vae_loss_mse(x.to(DEV), out, mu, logvar, c.kl_weight)
    assert abs(loss.item() - float(c.z["loss"])) <= 1e-2 * abs(float(c.z["loss"]))
    loss.backward()
    # gradients against the oracle (fp32, same masks and eps): direction and size
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o_out, o_mu, o_lv = O.unet_forward(c.variant, x, leaf, eps, True, c.p, masks)
    O.unet_loss_mse(x, o_out, o_mu, o_lv, c.kl_weight).backward()
    worst = 1.0
    tot_dot = tot_a = tot_b = 0.0
    for k, p in m.named_parameters():
        ga, gb = p.grad.detach().float().cpu().double().reshape(-1), leaf[k].grad.double().reshape(-1)
        assert torch.isfinite(ga).all(), k
        tot_dot += float(ga @ gb); tot_a += float(ga @ ga); tot_b += float(gb @ gb)
        if gb.norm() > 1e-6 * max(1.0, float(gb.numel()) ** 0.5):
            worst = min(worst, cosine(ga, gb))
            assert cosine(ga, gb) >= 0.9, (k, cosine(ga, gb))
            assert 0.8 <= float(ga.norm() / gb.norm()) <= 1.25, (k, float(ga.norm() / gb.norm()))
    assert tot_dot / (tot_a * tot_b) ** 0.5 >= 0.99
    print(f"{name}: worst per-tensor gradient cosine {worst:.4f}, overall {tot_dot / (tot_a * tot_b) ** 0.5:.5f}")


def test_unet_trains_with_philox_streams():
    """no injected eps / masks: the Philox streams drive dropout and eps, a repeated step index repeats the pass bit
    for bit, and a few AdamW steps on one batch bring the loss down"""
    from vae_amd.unet import VAEUnet256, vae_loss_mse
    torch.manual_seed(3)
    m = VAEUnet256(3, 64, 0.1, 32).to(DEV).train()
    x = torch.rand(8, 3, 32, 32, device=DEV)

    def rewind(step):
        m.rng_step = step
        for n_ in m.block_names():
            getattr(m, n_).rng_step = step
    rewind(5)
    a = [t.detach().clone() for t in m(x)]
    rewind(5)
    b = [t.detach().clone() for t in m(x)]
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    c = [t.detach().clone() for t in m(x)]             # next step index: other draws
    assert not torch.equal(a[0], c[0])
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    losses = []
    for _ in range(12):
        out, mu, logvar = m(x)
        loss = vae_loss_mse(x, out, mu, logvar, 0.1)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        losses.append(loss.item())
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < 0.8 * losses[0], losses


def test_draft_losses_against_their_formulas():
    """vae_loss_mse / vae_loss_bce of the draft script (vae/practice/vae_original.py:513-532) and their gradients"""
    from vae_amd.unet import vae_loss_bce, vae_loss_mse
    g = torch.Generator().manual_seed(5)
    x = torch.rand(6, 3, 16, 16, generator=g)
    logits = (4 * torch.randn(6, 3, 16, 16, generator=g)).requires_grad_(True)      # some beyond the +-10 clamp
    mu = torch.randn(6, 64, generator=g).requires_grad_(True)
    lv = (0.5 * torch.randn(6, 64, generator=g)).requires_grad_(True)
    kl = -0.5 * torch.sum(1 + lv - mu.pow(2) - lv.exp())
    refs = {"bce": F.binary_cross_entropy_with_logits(logits.clamp(-10, 10), x, reduction="mean") + kl,
            "mse": (F.mse_loss(logits, x, reduction="sum") + 0.3 * kl) / 6}
    for kind, ref in refs.items():
        gr = torch.autograd.grad(ref, (logits, mu, lv), retain_graph=True)
        d = [t.detach().to(DEV).requires_grad_(True) for t in (logits, mu, lv)]
        got = vae_loss_bce(x.to(DEV), *d) if kind == "bce" else vae_loss_mse(x.to(DEV), *d, 0.3)
        assert abs(got.item() - ref.item()) <= 1e-5 * abs(ref.item()), kind
        got.backward()
        for a, b in zip(d, gr):
            assert rel_l2(a.grad.cpu(), b) <= 1e-5, kind
