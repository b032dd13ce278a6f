"""The streaming GEMM kernel (first conv, ConvTranspose2d forward: resident weights, activations loaded
straight into the matrix-instruction operand registers) against torch on bf16-rounded operands, against the tiled
GEMM it replaces (same operands, fp32 accumulation in another order) and for its batch statistics, through the C ABI.
Geometries cover one pixel pair per wave, several pairs per wave (the prefetch path) and every template instance."""
import pytest
import torch
import torch.nn.functional as F

from golden_util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def L():
    from vae_amd import _lib
    _lib.load()
    yield _lib
    _lib.call("vae_set_tuning", b"stream_gemm", 0)


def bf(t):
    return t.to(torch.bfloat16).float()


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def from_nhwc(t):
    return t.float().cpu().permute(0, 3, 1, 2)


def check_bf16(got, ref, what):
    err = (got - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + 1e-3 * ref.abs().max()
    assert (err <= tol).all(), f"{what}: max err {err.max().item()} (ref max {ref.abs().max().item()})"
    assert rel_l2(got, ref) <= 3e-3, what


@pytest.mark.parametrize("n,H,W,Ci,Co", [(2, 8, 8, 3, 64), (3, 32, 32, 3, 64), (1, 4, 8, 1, 64), (96, 32, 32, 3, 64),
                                         (300, 16, 16, 3, 64), (5, 16, 16, 3, 128), (2, 8, 16, 8, 192)])
def test_first_conv_streaming_equals_torch_and_the_tiled_kernel(L, n, H, W, Ci, Co):
    g = torch.Generator().manual_seed(n + H)
    x = bf(torch.rand(n, Ci, H, W, generator=g))
    w = bf(torch.randn(Co, Ci, 3, 3, generator=g) / 5)
    b = torch.randn(Co, generator=g)
    ref = F.conv2d(x, w, b, padding=1)
    x8 = torch.zeros(n, H, W, 8, dtype=torch.bfloat16)
    x8[..., :Ci] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    x8 = x8.to(DEV)
    wp = torch.zeros(Co, 16, 8, dtype=torch.bfloat16)
    wp[:, :9, :Ci] = w.permute(0, 2, 3, 1).reshape(Co, 9, Ci).to(torch.bfloat16)
    wp = wp.to(DEV)
    bd = b.to(DEV)
    out = {}
    for mode in (0, 1):
        L.call("vae_set_tuning", b"stream_gemm", mode)
        rows = L.call("vae_conv3x3_c8_stat_rows", n, H, W, Co)
        assert 1 <= rows <= L.call("vae_gemm_stat_rows", n * H * W, Co)
        y = torch.full((n, H, W, Co), float("nan"), device=DEV, dtype=torch.bfloat16)
        ssum = torch.full((rows, Co), float("nan"), device=DEV)
        ssq = torch.full((rows, Co), float("nan"), device=DEV)
        L.call("vae_conv3x3_c8_gemm", L.ptr(x8), L.ptr(wp), L.ptr(bd), L.ptr(y), L.ptr(ssum), L.ptr(ssq), n, H, W, Co,
               L.current_stream())
        check_bf16(from_nhwc(y), ref, f"first conv, mode {mode}")
        assert rel_l2(ssum.sum(0).cpu(), ref.sum(dim=(0, 2, 3))) <= 1e-4
        assert rel_l2(ssq.sum(0).cpu(), (ref ** 2).sum(dim=(0, 2, 3))) <= 1e-4
        out[mode] = y.float().cpu()
    L.call("vae_set_tuning", b"stream_gemm", 0)
    # same operands, fp32 accumulation: the two kernels differ by summation order only (a bf16 ulp here and there)
    assert rel_l2(out[0], out[1]) <= 1e-3
    assert (out[0] != out[1]).float().mean().item() < 0.02


@pytest.mark.parametrize("n,H,W,Ci,Co", [(2, 8, 8, 3, 64), (3, 32, 32, 3, 64), (1, 4, 8, 1, 64), (96, 32, 32, 3, 64),
                                         (5, 16, 16, 3, 128), (2, 8, 16, 8, 256)])
def test_first_conv_in_two_passes_equals_conv_then_batchnorm(L, n, H, W, Ci, Co):
    """statistics-only pass == the statistics rows of the storing kernel, bit for bit; the BatchNorm + ReLU pass ==
    vae_bn_act_fwd applied to the stored conv output, bit for bit; the ReLU bits say where the activation is positive"""
    assert L.call("vae_conv3x3_c8_twice_ok", n, H, W, Co) == 1
    g = torch.Generator().manual_seed(n + H + 7)
    x8 = torch.zeros(n, H, W, 8, dtype=torch.bfloat16)
    x8[..., :Ci] = torch.rand(n, H, W, Ci, generator=g).to(torch.bfloat16)
    x8 = x8.to(DEV)
    wp = torch.zeros(Co, 16, 8, dtype=torch.bfloat16)
    wp[:, :9, :Ci] = (torch.randn(Co, 9, Ci, generator=g) / 5).to(torch.bfloat16)
    wp = wp.to(DEV)
    bd = torch.randn(Co, generator=g).to(DEV)
    rows = L.call("vae_conv3x3_c8_stat_rows", n, H, W, Co)
    y = torch.full((n, H, W, Co), float("nan"), device=DEV, dtype=torch.bfloat16)
    ssum, ssq = torch.full((rows, Co), float("nan"), device=DEV), torch.full((rows, Co), float("nan"), device=DEV)
    s = L.current_stream()
    L.call("vae_conv3x3_c8_gemm", L.ptr(x8), L.ptr(wp), L.ptr(bd), L.ptr(y), L.ptr(ssum), L.ptr(ssq), n, H, W, Co, s)
    s1, q1 = torch.full_like(ssum, float("nan")), torch.full_like(ssq, float("nan"))
    L.call("vae_conv3x3_c8_stats", L.ptr(x8), L.ptr(wp), L.ptr(bd), L.ptr(s1), L.ptr(q1), n, H, W, Co, s)
    assert torch.equal(s1, ssum) and torch.equal(q1, ssq)
    gam, bet = (torch.rand(Co, generator=g) + 0.5).to(DEV), (torch.randn(Co, generator=g) / 3).to(DEV)
    coef = torch.empty(4, Co, device=DEV)
    L.call("vae_bn_finalize", L.ptr(ssum), L.ptr(ssq), rows, Co, 1, float(n * H * W), L.ptr(gam), L.ptr(bet), None, None, 0.1, 1e-5, 0,
           L.ptr(coef[0]), L.ptr(coef[1]), L.ptr(coef[2]), L.ptr(coef[3]), s)
    ref = torch.full_like(y, float("nan"))
    L.call("vae_bn_act_fwd", L.ptr(y), L.ptr(coef[0]), L.ptr(coef[1]), L.ptr(ref), n, H, W, Co, 0, 0.0, None, 0, 0, 0, 0, s)
    out = torch.full_like(y, float("nan"))
    bits = torch.full((n * H * W, Co // 8), 0xAA, device=DEV, dtype=torch.uint8)
    L.call("vae_conv3x3_c8_bn_act", L.ptr(x8), L.ptr(wp), L.ptr(bd), L.ptr(coef[0]), L.ptr(coef[1]), L.ptr(out), L.ptr(bits),
           n, H, W, Co, s)
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    want = (ref.float().reshape(n * H * W, Co // 8, 8) > 0).to(torch.int32)
    packed = (want << torch.arange(8, device=DEV, dtype=torch.int32)).sum(-1).to(torch.uint8)
    # (an activation that is positive in fp32 and rounds to a positive bf16: the bit and the stored sign agree)
    assert torch.equal(bits, packed)
    # without the bits
    out2 = torch.full_like(y, float("nan"))
    L.call("vae_conv3x3_c8_bn_act", L.ptr(x8), L.ptr(wp), L.ptr(bd), L.ptr(coef[0]), L.ptr(coef[1]), L.ptr(out2), None, n, H, W, Co, s)
    assert torch.equal(out2.view(torch.int16), out.view(torch.int16))


def test_first_conv_in_two_passes_rejects(L):
    assert L.call("vae_conv3x3_c8_twice_ok", 1, 6, 6, 64) == 0        # not a power of two
    assert L.call("vae_conv3x3_c8_twice_ok", 1, 8, 8, 48) == 0        # channels
    t = torch.zeros(4096, device=DEV)
    with pytest.raises(RuntimeError, match="not covered"):
        L.call("vae_conv3x3_c8_stats", L.ptr(t), L.ptr(t), None, L.ptr(t), L.ptr(t), 1, 6, 6, 64, L.current_stream())
    with pytest.raises(RuntimeError, match="null pointer"):
        L.call("vae_conv3x3_c8_bn_act", L.ptr(t), L.ptr(t), None, None, None, L.ptr(t), None, 1, 8, 8, 64, L.current_stream())


@pytest.mark.parametrize("n,H,W,Ci,Co", [(2, 4, 4, 256, 128), (3, 8, 8, 128, 64), (2, 16, 16, 64, 64), (128, 16, 16, 64, 64),
                                         (512, 8, 8, 128, 64), (700, 4, 4, 256, 128), (400, 16, 16, 64, 64),
                                         (2, 8, 8, 64, 128), (64, 8, 8, 256, 64)])
def test_convt2x2_streaming_fwd(L, n, H, W, Ci, Co):
    g = torch.Generator().manual_seed(n + H + Ci)
    x = bf(torch.randn(n, Ci, H, W, generator=g))
    w = bf(torch.randn(Ci, Co, 2, 2, generator=g) / Ci ** 0.5)
    b = torch.randn(Co, generator=g)
    ref = F.conv_transpose2d(x, w, b, stride=2)
    wp = w.permute(2, 3, 1, 0).contiguous().to(torch.bfloat16).to(DEV)      # [(kh,kw,co)][ci]
    wd = w.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)      # [ci][(kh,kw,co)]
    dy = bf(torch.randn(n, Co, 2 * H, 2 * W, generator=g))
    xr = x.clone().requires_grad_(True)
    F.conv_transpose2d(xr, w, None, stride=2).backward(dy)
    xd, bd, dyd = nhwc(x), b.to(DEV), nhwc(dy)
    outs = {}
    for mode in (2, 1):
        L.call("vae_set_tuning", b"stream_gemm", mode)
        rows = L.call("vae_convt2x2_stat_rows", n, H, W, Ci, Co)
        assert 1 <= rows <= L.call("vae_gemm_stat_rows", n * H * W, 4 * Co)
        y = torch.full((n, 2 * H, 2 * W, Co), float("nan"), device=DEV, dtype=torch.bfloat16)
        fold = L.call("vae_convt2x2_stat_fold", n, H, W, Ci, Co)
        assert fold in (1, 2, 4)
        ssum = torch.full((rows, fold * Co), float("nan"), device=DEV)
        ssq = torch.full((rows, fold * Co), float("nan"), device=DEV)
        L.call("vae_convt2x2_fwd_gemm", L.ptr(xd), L.ptr(wp), L.ptr(bd), L.ptr(y), L.ptr(ssum), L.ptr(ssq), n, H, W, Ci, Co,
               L.current_stream())
        check_bf16(from_nhwc(y), ref, f"convT fwd, mode {mode}")
        assert rel_l2(ssum.sum(0).view(fold, Co).sum(0).cpu(), ref.sum(dim=(0, 2, 3))) <= 1e-4
        assert rel_l2(ssq.sum(0).view(fold, Co).sum(0).cpu(), (ref ** 2).sum(dim=(0, 2, 3))) <= 1e-4
        # without statistics
        y2 = torch.full_like(y, float("nan"))
        L.call("vae_convt2x2_fwd_gemm", L.ptr(xd), L.ptr(wp), L.ptr(bd), L.ptr(y2), None, None, n, H, W, Ci, Co, L.current_stream())
        assert torch.equal(y, y2)
        dx = torch.full((n, H, W, Ci), float("nan"), device=DEV, dtype=torch.bfloat16)
        L.call("vae_convt2x2_dgrad_gemm", L.ptr(dyd), L.ptr(wd), L.ptr(dx), n, H, W, Ci, Co, L.current_stream())
        check_bf16(from_nhwc(dx), xr.grad, f"convT dgrad, mode {mode}")
        outs[mode] = (y.float().cpu(), dx.float().cpu())
    L.call("vae_set_tuning", b"stream_gemm", 0)
    assert rel_l2(outs[2][0], outs[1][0]) <= 1e-3 and torch.equal(outs[2][1], outs[1][1])


@pytest.mark.parametrize("n,H,W", [(2, 8, 8), (16, 16, 16), (3, 4, 16), (1, 8, 8), (40, 16, 16), (2, 6, 16), (4, 12, 12)])
def test_convt_backward_in_one_kernel(L, n, H, W):
    """ConvTranspose2d(64, 64, 2, 2) backward (vae/models/vae.py:131): input gradient and weight gradient from one read of
    dY, against torch autograd on the bf16-rounded operands and against the two separate kernels.  Sizes: one tile, many
    tiles per workgroup (40 * 256 pixels = 160 tiles... on up to 256 workgroups), a non-square grid."""
    Ci = Co = 64
    assert L.call("vae_convt2x2_bwd_fused_ok", n, H, W, Ci, Co) == 1
    assert L.call("vae_convt2x2_bwd_fused_ok", n, H, W, 128, 64) == 0
    g = torch.Generator().manual_seed(n * 100 + H)
    x = bf(torch.randn(n, Ci, H, W, generator=g))
    w = bf(torch.randn(Ci, Co, 2, 2, generator=g) / Ci ** 0.5)
    dy = bf(torch.randn(n, Co, 2 * H, 2 * W, generator=g))
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    F.conv_transpose2d(xr, wr, None, stride=2).backward(dy)
    wd = w.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)      # [ci][(kh,kw,co)]
    xd, dyd = nhwc(x), nhwc(dy)
    ns = L.call("vae_convt2x2_bwd_fused_slabs", n, H, W)
    assert 1 <= ns <= 256
    dx = torch.full((n, H, W, Ci), float("nan"), device=DEV, dtype=torch.bfloat16)
    slab = torch.full((ns, Co, 4, Ci), float("nan"), device=DEV)
    L.call("vae_convt2x2_bwd_fused", L.ptr(dyd), L.ptr(xd), L.ptr(wd), L.ptr(dx), L.ptr(slab), n, H, W, Ci, Co, L.current_stream())
    check_bf16(from_nhwc(dx), xr.grad, "fused convT dgrad")
    dw = slab.sum(0).permute(2, 0, 1).reshape(Ci, Co, 2, 2).cpu()          # [co][tap][ci] -> [ci][co][kh][kw]
    assert rel_l2(dw, wr.grad) <= 1e-5
    # the separate kernels
    dx2 = torch.full_like(dx, float("nan"))
    L.call("vae_convt2x2_dgrad_gemm", L.ptr(dyd), L.ptr(wd), L.ptr(dx2), n, H, W, Ci, Co, L.current_stream())
    assert rel_l2(dx.float().cpu(), dx2.float().cpu()) <= 1e-3
    M = n * H * W
    ns2 = L.call("vae_wgrad_ksplit", Co, Ci, 4, M)
    slab2 = torch.full((ns2, Co, 4, Ci), float("nan"), device=DEV)
    L.call("vae_wgrad_gemm", L.ptr(dyd), L.ptr(xd), L.ptr(slab2), Co, Ci, M, Co, Ci, 2, H, W, ns2, L.current_stream())
    assert rel_l2(slab.sum(0).cpu(), slab2.sum(0).cpu()) <= 1e-5
    with pytest.raises(RuntimeError, match="64 -> 64 channels only"):
        L.call("vae_convt2x2_bwd_fused", L.ptr(dyd), L.ptr(xd), L.ptr(wd), L.ptr(dx), L.ptr(slab), n, H, W, 128, 64, L.current_stream())


# the last case: 640 tiles over 256 workgroups, two or three tiles each (odd and even pass counts, the prefetch ring wrapping)
@pytest.mark.parametrize("n,H,W,Co_head", [(2, 8, 8, 3), (24, 16, 16, 3), (1, 8, 8, 1), (3, 4, 16, 4), (40, 32, 32, 3),
                                                   (2, 6, 16, 3), (4, 12, 12, 2)])       # the last two: sizes that are no powers of two
def test_head_and_convt_backward_without_the_gradient_tensor(L, n, H, W, Co_head):
    """bn1 + ReLU + 1x1 head backward followed by the last ConvTranspose's backward (vae/models/vae.py:131-134): the pair
    vae_head_bwd_sums + vae_head_convt_bwd_fused computes the head's gradient tile by tile inside the ConvTranspose kernel
    and must give what vae_head_bwd (gradient stored as bf16) + vae_convt2x2_bwd_fused give, to the bf16 rounding of that
    gradient, and what an fp64 evaluation of the formulas gives."""
    g = torch.Generator().manual_seed(n + H + Co_head)
    C = 64
    y = bf(torch.randn(n, 2 * H, 2 * W, C, generator=g) * 1.3 + 0.2).to(torch.bfloat16).to(DEV)      # ConvT output, NHWC
    x = bf(torch.randn(n, H, W, C, generator=g)).to(torch.bfloat16).to(DEV)
    wct = bf(torch.randn(C, C, 2, 2, generator=g) / 8)
    wd = wct.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    yf = y.float()
    mean, var = yf.mean((0, 1, 2)), yf.var((0, 1, 2), unbiased=False)
    rstd = 1 / torch.sqrt(var + 1e-5)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.2).to(DEV)
    sc = gamma * rstd
    sh = beta - mean * sc
    wh = (torch.randn(Co_head, C, generator=g) * 0.2).to(DEV)
    recon = torch.rand(n, Co_head, 2 * H, 2 * W, generator=g).to(DEV) * 0.9 + 0.05
    drecon = torch.randn(n, Co_head, 2 * H, 2 * W, generator=g).to(DEV) * 1e-3
    s = L.current_stream()
    nb = L.call("vae_head_bwd_blocks", n, 2 * H, 2 * W)
    ncols = (3 + Co_head) * 64
    # stored form
    part = torch.empty(nb, ncols, device=DEV)
    sums1, sh1 = torch.empty(128, device=DEV), torch.empty(Co_head * 65, device=DEV)
    gbuf = torch.full((n, 2 * H, 2 * W, C), float("nan"), device=DEV, dtype=torch.bfloat16)
    L.call("vae_head_bwd", L.ptr(drecon), L.ptr(recon), L.ptr(y), L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(rstd), L.ptr(wh),
           L.ptr(part), L.ptr(sums1), L.ptr(sh1), L.ptr(gbuf), n, 2 * H, 2 * W, Co_head, s)
    ns = L.call("vae_convt2x2_bwd_fused_slabs", n, H, W)
    dx1 = torch.full((n, H, W, C), float("nan"), device=DEV, dtype=torch.bfloat16)
    slab1 = torch.full((ns, C, 4, C), float("nan"), device=DEV)
    L.call("vae_convt2x2_bwd_fused", L.ptr(gbuf), L.ptr(x), L.ptr(wd), L.ptr(dx1), L.ptr(slab1), n, H, W, C, C, s)
    # on-the-fly form
    part2 = torch.empty(nb, ncols, device=DEV)
    sums2, sh2 = torch.empty(128, device=DEV), torch.empty(Co_head * 65, device=DEV)
    dl = torch.full((n * 4 * H * W, 4), float("nan"), device=DEV)
    L.call("vae_head_bwd_sums", L.ptr(drecon), L.ptr(recon), L.ptr(y), L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(rstd), L.ptr(wh),
           L.ptr(part2), L.ptr(sums2), L.ptr(sh2), L.ptr(dl), n, 2 * H, 2 * W, Co_head, s)
    assert torch.equal(sums1, sums2) and torch.equal(sh1, sh2)
    dl_ref = (drecon * recon * (1 - recon)).permute(0, 2, 3, 1).reshape(-1, Co_head)
    assert torch.equal(dl[:, :Co_head], dl_ref) and float(dl[:, Co_head:].abs().max() if Co_head < 4 else 0) == 0
    dx2 = torch.full_like(dx1, float("nan"))
    slab2 = torch.full_like(slab1, float("nan"))
    L.call("vae_head_convt_bwd_fused", L.ptr(y), L.ptr(dl), L.ptr(sc), L.ptr(sh), L.ptr(mean), L.ptr(rstd), L.ptr(sums2), L.ptr(wh),
           Co_head, L.ptr(x), L.ptr(wd), L.ptr(dx2), L.ptr(slab2), n, H, W, s)
    # the gradient is formed as one affine expression here and in head_bwd's own order there: bf16 roundings of it differ
    # in the last bit now and then
    assert rel_l2(dx2.float().cpu(), dx1.float().cpu()) <= 2e-3
    assert rel_l2(slab2.sum(0).cpu(), slab1.sum(0).cpu()) <= 2e-3
    # and against fp64: the head's gradient, then the ConvTranspose's two gradients
    v = lambda t: t.double().cpu()
    yd_, dlr = v(yf), v(dl_ref).view(n, 2 * H, 2 * W, Co_head)
    pos = (yf * sc + sh > 0).cpu()
    da = torch.einsum("nhwo,oc->nhwc", dlr, v(wh))
    gg = torch.where(pos, da, torch.zeros((), dtype=torch.float64))
    xh = (yd_ - v(mean)) * v(rstd)
    cnt = n * 4 * H * W
    gref = v(sc) * (gg - gg.sum((0, 1, 2)) / cnt - xh * (gg * xh).sum((0, 1, 2)) / cnt)
    xr, wr = v(x.float()).permute(0, 3, 1, 2).requires_grad_(True), wct.double().requires_grad_(True)
    F.conv_transpose2d(xr, wr, None, stride=2).backward(gref.permute(0, 3, 1, 2))
    assert rel_l2(dx2.float().cpu().permute(0, 3, 1, 2), xr.grad.float()) <= 4e-3          # g rounded to bf16 in LDS, dx to bf16
    assert rel_l2(slab2.sum(0).permute(2, 0, 1).reshape(C, C, 2, 2).cpu(), wr.grad.float()) <= 4e-3
