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


@pytest.mark.parametrize("n,H,W", [(2, 8, 8), (16, 16, 16), (3, 4, 16), (1, 8, 8), (40, 16, 16)])
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
