"""Per-kernel parity of the HIP path (through the C ABI) against the CPU oracle.

Tolerances: fp32 kernels 1e-5 relative (per-tensor L2); bf16 MFMA kernels are compared with
the oracle evaluated on the SAME bf16-rounded operands, so the only differences are fp32
summation order and the final bf16 rounding of the output (<= 2^-8 relative per element,
~1.2e-3 rms) -- far inside the 2e-2 bf16 budget, and tight enough to catch indexing bugs.
RNG streams (uint32 words, keep-masks, normals) are compared bit for bit.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vae_oracle as O
from golden_util import rel_l2

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def L():
    from vae_amd import _lib
    _lib.load()
    return _lib


def bf(t):
    return t.to(torch.bfloat16).float()


def nhwc_bf16(t):
    return t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def from_nhwc(t):
    return t.float().cpu().permute(0, 3, 1, 2)


# ------------------------------------------------------------------ RNG
@pytest.mark.parametrize("n", [1, 5, 4096, 100003])
def test_philox_streams_bit_exact(L, n):
    seed, step, stream, rank = 0x1234567890ABCDEF, 7, 4, 3
    out = torch.empty(n, dtype=torch.int32, device=DEV)
    L.call("vae_philox_fill", L.ptr(out), n, 0, 0.0, seed, step, stream, rank, L.current_stream())
    ref = O.philox_bits(n, seed, stream, step, rank)
    assert (out.cpu().numpy().view(np.uint32) == ref).all()
    z = torch.empty(n, dtype=torch.float32, device=DEV)
    L.call("vae_philox_fill", L.ptr(z), n, 1, 0.0, seed, step, stream, rank, L.current_stream())
    refz = O.philox_normal(n, seed, stream, step, rank)
    assert (z.cpu().numpy().view(np.uint32) == refz.view(np.uint32)).all(), "normals not bit-exact"
    m = torch.empty(n, dtype=torch.uint8, device=DEV)
    L.call("vae_philox_fill", L.ptr(m), n, 2, 0.1, seed, step, 1, rank, L.current_stream())
    refm = O.philox_keep_mask(n, 0.1, seed, 1, step, rank)
    assert (m.cpu().numpy() == refm).all()
    assert L.call("vae_dropout_threshold", 0.1) == O.dropout_threshold(0.1)


def test_generator_logarithm_bit_exact_over_its_whole_domain(L):
    """the normal generator takes ln(u) of the 2^24 uniforms (k + 1) 2^-24 and of nothing else: every one of them
    against the host restatement (numpy's correctly rounded fp32 division inside), bit for bit -- the kernel's
    division is a reciprocal + one correction, which is only as good as this table says"""
    n = 1 << 24
    out = torch.empty(n, dtype=torch.float32, device=DEV)
    L.call("vae_philox_fill", L.ptr(out), n, 3, 0.0, 0, 0, 0, 0, L.current_stream())
    u = ((np.arange(n, dtype=np.float32) + np.float32(1.0)) * np.float32(2.0 ** -24)).astype(np.float32)
    ref = O._log_f32(u)
    got = out.cpu().numpy()
    assert (got.view(np.uint32) == ref.view(np.uint32)).all(), int((got.view(np.uint32) != ref.view(np.uint32)).sum())


# ------------------------------------------------------------------ loss
@pytest.mark.parametrize("kind", list(range(5)))
@pytest.mark.parametrize("n", [4 * 3 * 32 * 32, 1027])
def test_recon_loss_fwd_bwd(L, kind, n):
    g = torch.Generator().manual_seed(kind * 10 + 1)
    if O.RECON_KINDS[kind].startswith("bce_logits"):
        # logits beyond +-10: the clamped draft form stops there (no gradient outside), the torch callable's
        # own form (kind 4) does not
        recon = (torch.rand(n, generator=g) * 30 - 15)
    else:
        recon = torch.rand(n, generator=g) * 0.98 + 0.01
    target = torch.rand(n, generator=g)
    recon[:3] = target[:3]                                      # exact ties (sign(0) = 0 for L1)
    r = recon.clone().requires_grad_(True)
    ref = O.recon_loss_sum(r, target, O.RECON_KINDS[kind])
    ref.backward()
    rd, td = recon.to(DEV), target.to(DEV)
    grid = L.call("vae_recon_loss_grid", n)
    partial = torch.zeros(grid, device=DEV)
    grad = torch.empty(n, device=DEV)
    scale = 0.25
    L.call("vae_recon_loss_fwd_bwd", L.ptr(rd), L.ptr(td), L.ptr(grad), L.ptr(partial), n, kind, scale,
           L.current_stream())
    out3 = torch.zeros(3, device=DEV)
    L.call("vae_elbo_combine", L.ptr(partial), grid, None, 0, 1.0, 1.0, L.ptr(out3), L.current_stream())
    got = out3.cpu()
    assert abs(got[1].item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert rel_l2(grad.cpu(), r.grad * scale) <= 1e-5
    # forward-only variant leaves grad untouched and gives the same sum
    partial2 = torch.zeros(grid, device=DEV)
    L.call("vae_recon_loss_fwd_bwd", L.ptr(rd), L.ptr(td), None, L.ptr(partial2), n, kind, scale,
           L.current_stream())
    assert torch.equal(partial, partial2)


def test_recon_loss_rejects_bad_args(L):
    x = torch.zeros(8, device=DEV)
    with pytest.raises(RuntimeError, match="n_elems"):
        L.call("vae_recon_loss_fwd_bwd", L.ptr(x), L.ptr(x), None, L.ptr(x), 0, 0, 1.0, L.current_stream())
    with pytest.raises(RuntimeError, match="kind"):
        L.call("vae_recon_loss_fwd_bwd", L.ptr(x), L.ptr(x), None, L.ptr(x), 8, 9, 1.0, L.current_stream())


# ------------------------------------------------------------------ reparam + KL
@pytest.mark.parametrize("n,lat", [(4, 128), (3, 32), (512, 128)])
@pytest.mark.parametrize("inject", [True, False])
def test_reparam_kl(L, n, lat, inject):
    g = torch.Generator().manual_seed(n * 7 + lat)
    mu = torch.randn(n, lat, generator=g)
    logvar = torch.randn(n, lat, generator=g) * 0.5
    seed, step, rank = 1337, 5, 2
    if inject:
        eps = torch.randn(n, lat, generator=g)
    else:
        eps = torch.from_numpy(O.philox_normal(n * lat, seed, O.STREAM_EPS, step, rank)).view(n, lat)
    mu_r, lv_r = mu.clone().requires_grad_(True), logvar.clone().requires_grad_(True)
    z_ref = O.reparameterize(mu_r, lv_r, eps)
    kl_ref = O.kl_sum(mu_r, lv_r)
    dz = torch.randn(n, lat, generator=g)
    kl_scale = 4.0 / n
    ((z_ref * dz).sum() + kl_scale * kl_ref).backward()

    mud, lvd = mu.to(DEV), logvar.to(DEV)
    epsd = eps.to(DEV) if inject else None
    ne = n * lat
    grid = L.call("vae_reparam_grid", ne)
    z = torch.empty(n, lat, device=DEV)
    zb = torch.empty(n, lat, device=DEV, dtype=torch.bfloat16)
    eo = torch.empty(n, lat, device=DEV)
    klp = torch.zeros(grid, device=DEV)
    L.call("vae_reparam_kl_fwd", L.ptr(mud), L.ptr(lvd), L.ptr(epsd), L.ptr(z), L.ptr(zb), L.ptr(eo),
           L.ptr(klp), ne, seed, step, rank, L.current_stream())
    assert torch.equal(eo.cpu(), eps), "eps stream differs"
    assert rel_l2(z.cpu(), z_ref.detach()) <= 1e-5
    assert rel_l2(zb.float().cpu(), z_ref.detach()) <= 4e-3
    assert abs(klp.sum().item() - kl_ref.item()) <= 1e-5 * abs(kl_ref.item())
    dmu = torch.empty(n, lat, device=DEV)
    dlv = torch.empty(n, lat, device=DEV)
    dcat = torch.empty(n, 2 * lat, device=DEV, dtype=torch.bfloat16)
    dzd = dz.to(DEV)
    L.call("vae_reparam_kl_bwd", L.ptr(dzd), L.ptr(mud), L.ptr(lvd), L.ptr(epsd), None, None,
           kl_scale, L.ptr(dmu), L.ptr(dlv), L.ptr(dcat), ne, lat, seed, step, rank, L.current_stream())
    assert rel_l2(dmu.cpu(), mu_r.grad) <= 1e-5
    assert rel_l2(dlv.cpu(), lv_r.grad) <= 1e-5
    assert rel_l2(dcat.float().cpu(), torch.cat([mu_r.grad, lv_r.grad], 1)) <= 4e-3


# ------------------------------------------------------------------ GEMM layers
def pack_conv3(w):          # [Co,Ci,3,3] -> [Co][tap][Ci]
    return w.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def pack_conv3_dgrad(w):    # [Co,Ci,3,3] -> [Ci][flipped tap][Co]
    return w.flip(2, 3).permute(1, 2, 3, 0).contiguous().to(torch.bfloat16).to(DEV)


def check_bf16(got, ref, what):
    err = (got - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + 1e-3 * ref.abs().max()
    assert (err <= tol).all(), f"{what}: max err {err.max().item()} (ref max {ref.abs().max().item()})"
    assert rel_l2(got, ref) <= 3e-3, what


# the last two cases are large enough (>= 256 workgroups) to run the 128-row tiles, the others
# run the 64-row tiles chosen for small problems
@pytest.mark.parametrize("n,H,W,Ci,Co", [(2, 4, 4, 64, 64), (3, 8, 8, 64, 128), (2, 16, 16, 128, 128),
                                         (5, 4, 4, 256, 256), (2, 32, 32, 64, 64), (1, 6, 10, 128, 64),
                                         (128, 16, 16, 128, 128), (64, 32, 32, 64, 64),
                                         # patch-staged kernel geometries: 8 / 2 / 4 images per tile, a ragged last tile,
                                         # 64-wide rows, several channel chunks with 256 outputs
                                         (2048, 4, 4, 64, 128), (513, 8, 8, 128, 64), (1024, 8, 4, 64, 64),
                                         (8, 64, 64, 64, 64), (40, 32, 32, 128, 256),
                                         # enough 64-channel work for the 256-pixel tiles: four 8x8 images, one 16x16 image,
                                         # 8 rows of 32, 4 rows of 64 per tile; two channel chunks
                                         (2048, 8, 8, 64, 64), (512, 16, 16, 128, 64), (128, 32, 32, 64, 64), (32, 64, 64, 64, 64),
                                         # channel counts that are multiples of 64 but not powers of two (concatenated
                                         # skip inputs), on the small-problem tiles and on the patch kernel
                                         (4, 16, 16, 192, 64), (4, 16, 16, 64, 192), (3, 8, 8, 384, 128), (3, 8, 8, 128, 384),
                                         (300, 16, 16, 192, 64), (300, 16, 16, 64, 192)])
def test_conv3x3_fwd_dgrad(L, n, H, W, Ci, Co):
    g = torch.Generator().manual_seed(n + H + Ci)
    x = bf(torch.randn(n, Ci, H, W, generator=g))
    w = bf(torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5))
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    M = n * H * W
    gm = L.call("vae_conv3x3_stat_rows", n, H, W, Ci, Co)
    assert 1 <= gm <= L.call("vae_gemm_stat_rows", M, Co)
    assert gm <= (M + L.call("vae_gemm_rows_per_block") - 1) // L.call("vae_gemm_rows_per_block")
    y = torch.full((n, H, W, Co), float("nan"), device=DEV, dtype=torch.bfloat16)
    ssum = torch.full((gm, Co), float("nan"), device=DEV)
    ssq = torch.full((gm, Co), float("nan"), device=DEV)
    xd, wd, bd = nhwc_bf16(x), pack_conv3(w), b.to(DEV)      # keep the device tensors alive
    L.call("vae_conv3x3_gemm", L.ptr(xd), L.ptr(wd), L.ptr(bd), L.ptr(y),
           L.ptr(ssum), L.ptr(ssq), n, H, W, Ci, Co, L.current_stream())
    check_bf16(from_nhwc(y), ref, "conv3x3 fwd")
    assert rel_l2(ssum.sum(0).cpu(), ref.sum(dim=(0, 2, 3))) <= 1e-4
    assert rel_l2(ssq.sum(0).cpu(), (ref ** 2).sum(dim=(0, 2, 3))) <= 1e-4
    # dgrad: same kernel, flipped/transposed pack, no bias, no stats
    dy = bf(torch.randn(n, Co, H, W, generator=g))
    xr = x.clone().requires_grad_(True)
    F.conv2d(xr, w, None, padding=1).backward(dy)
    dx = torch.full((n, H, W, Ci), float("nan"), device=DEV, dtype=torch.bfloat16)
    dyd, wdd = nhwc_bf16(dy), pack_conv3_dgrad(w)
    L.call("vae_conv3x3_gemm", L.ptr(dyd), L.ptr(wdd), None, L.ptr(dx), None, None,
           n, H, W, Co, Ci, L.current_stream())
    check_bf16(from_nhwc(dx), xr.grad, "conv3x3 dgrad")


@pytest.mark.parametrize("n,H,W,Ci,Co", [(2, 4, 4, 256, 128), (3, 8, 8, 128, 64), (2, 16, 16, 64, 64), (1, 3, 5, 64, 64),
                                         (128, 16, 16, 64, 64)])
def test_convt2x2_fwd_dgrad(L, n, H, W, Ci, Co):
    g = torch.Generator().manual_seed(n + H + Ci)
    x = bf(torch.randn(n, Ci, H, W, generator=g))
    w = bf(torch.randn(Ci, Co, 2, 2, generator=g) / Ci ** 0.5)
    b = torch.randn(Co, generator=g)
    ref = F.conv_transpose2d(x, w, b, stride=2)
    wp = w.permute(2, 3, 1, 0).contiguous().to(torch.bfloat16).to(DEV)      # [(kh,kw,co)][ci]
    gm = L.call("vae_convt2x2_stat_rows", n, H, W, Ci, Co)
    assert 1 <= gm <= L.call("vae_gemm_stat_rows", n * H * W, 4 * Co)
    y = torch.full((n, 2 * H, 2 * W, Co), float("nan"), device=DEV, dtype=torch.bfloat16)
    fold = L.call("vae_convt2x2_stat_fold", n, H, W, Ci, Co)
    ssum = torch.full((gm, fold * Co), float("nan"), device=DEV)
    ssq = torch.full((gm, fold * Co), float("nan"), device=DEV)
    xd, bd = nhwc_bf16(x), b.to(DEV)
    L.call("vae_convt2x2_fwd_gemm", L.ptr(xd), L.ptr(wp), L.ptr(bd), L.ptr(y), L.ptr(ssum),
           L.ptr(ssq), n, H, W, Ci, Co, L.current_stream())
    check_bf16(from_nhwc(y), ref, "convT fwd")
    assert rel_l2(ssum.sum(0).view(fold, Co).sum(0).cpu(), ref.sum(dim=(0, 2, 3))) <= 1e-4
    assert rel_l2(ssq.sum(0).view(fold, Co).sum(0).cpu(), (ref ** 2).sum(dim=(0, 2, 3))) <= 1e-4
    dy = bf(torch.randn(n, Co, 2 * H, 2 * W, generator=g))
    xr = x.clone().requires_grad_(True)
    F.conv_transpose2d(xr, w, None, stride=2).backward(dy)
    wd = w.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)      # [ci][(kh,kw,co)]
    dx = torch.full((n, H, W, Ci), float("nan"), device=DEV, dtype=torch.bfloat16)
    dyd = nhwc_bf16(dy)
    L.call("vae_convt2x2_dgrad_gemm", L.ptr(dyd), L.ptr(wd), L.ptr(dx), n, H, W, Ci, Co,
           L.current_stream())
    check_bf16(from_nhwc(dx), xr.grad, "convT dgrad")


@pytest.mark.parametrize("M,N,K", [(4, 256, 4096), (130, 64, 128), (512, 4096, 128), (7, 1024, 64), (300, 64, 32)])
def test_linear_gemm_epilogues(L, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    a = bf(torch.randn(M, K, generator=g))
    w = bf(torch.randn(N, K, generator=g) / K ** 0.5)
    b = torch.randn(N, generator=g)
    ref = F.linear(a, w, b)
    ad, wd, bd = a.to(torch.bfloat16).to(DEV), w.to(torch.bfloat16).to(DEV), b.to(DEV)
    s = L.current_stream()
    out = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    L.call("vae_linear_gemm", L.ptr(ad), L.ptr(wd), L.ptr(bd), L.ptr(out), None, None, None, M, N, K, 0, s)
    check_bf16(out.float().cpu(), ref, "linear bf16")
    if K % 64:
        return
    out = torch.full((M, N), float("nan"), device=DEV, dtype=torch.bfloat16)
    L.call("vae_linear_gemm", L.ptr(ad), L.ptr(wd), L.ptr(bd), L.ptr(out), None, None, None, M, N, K, 1, s)
    check_bf16(out.float().cpu(), torch.tanh(ref), "linear tanh")
    o32 = torch.full((M, N), float("nan"), device=DEV)
    L.call("vae_linear_gemm", L.ptr(ad), L.ptr(wd), None, L.ptr(o32), None, None, None, M, N, K, 2, s)
    assert rel_l2(o32.cpu(), F.linear(a, w)) <= 1e-5
    o1 = torch.full((M, N // 2), float("nan"), device=DEV)
    o2 = torch.full((M, N // 2), float("nan"), device=DEV)
    L.call("vae_linear_gemm", L.ptr(ad), L.ptr(wd), L.ptr(bd), L.ptr(o1), L.ptr(o2), None, None, M, N, K, 3, s)
    assert rel_l2(torch.cat([o1, o2], 1).cpu(), ref) <= 1e-5


def test_gemm_rejects_bad_shapes(L):
    x = torch.zeros(1024, device=DEV, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        L.call("vae_conv3x3_gemm", L.ptr(x), L.ptr(x), None, L.ptr(x), None, None, 1, 2, 2, 3, 64, L.current_stream())
    with pytest.raises(RuntimeError, match="K="):
        L.call("vae_linear_gemm", L.ptr(x), L.ptr(x), None, L.ptr(x), None, None, None, 4, 64, 48, 0, L.current_stream())
