"""nn.MaxPool2d(2) on its own (NHWC bf16): forward, route bytes, backward from the route bytes and recomputed from the
input, against torch's max_pool2d and its autograd."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def L():
    from vae_amd import _lib
    _lib.load()
    return _lib


@pytest.mark.parametrize("n,H,W,C", [(3, 8, 8, 64), (2, 6, 10, 128), (5, 2, 2, 256), (1, 32, 32, 64), (7, 4, 12, 8)])
def test_maxpool2_fwd_bwd(L, n, H, W, C):
    g = torch.Generator().manual_seed(n + H + C)
    x = torch.randn(n, C, H, W, generator=g).to(torch.bfloat16).float()
    x[:, :, ::2, ::2] = torch.maximum(x[:, :, ::2, ::2], x[:, :, 1::2, 1::2])      # plenty of exact ties: the first maximum must win
    xr = x.clone().requires_grad_(True)
    ref = F.max_pool2d(xr, 2)
    gout = torch.randn(ref.shape, generator=g).to(torch.bfloat16).float()
    ref.backward(gout)
    xd = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    out = torch.full((n, H // 2, W // 2, C), float("nan"), device=DEV, dtype=torch.bfloat16)
    route = torch.full((n, H // 2, W // 2, C), 255, device=DEV, dtype=torch.uint8)
    L.call("vae_maxpool2_fwd", L.ptr(xd), L.ptr(out), L.ptr(route), n, H, W, C, L.current_stream())
    assert torch.equal(out.float().cpu().permute(0, 3, 1, 2), ref.detach())
    assert int(route.max()) <= 3
    gd = gout.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)
    for use_route in (True, False):
        dx = torch.full((n, H, W, C), float("nan"), device=DEV, dtype=torch.bfloat16)
        L.call("vae_maxpool2_bwd", L.ptr(gd), None if use_route else L.ptr(xd), L.ptr(route) if use_route else None, L.ptr(dx),
               n, H, W, C, L.current_stream())
        assert torch.equal(dx.float().cpu().permute(0, 3, 1, 2), xr.grad), use_route
    out2 = torch.empty_like(out)
    L.call("vae_maxpool2_fwd", L.ptr(xd), L.ptr(out2), None, n, H, W, C, L.current_stream())      # route is optional
    assert torch.equal(out2, out)
    with pytest.raises(RuntimeError, match="pooling needs even"):
        L.call("vae_maxpool2_fwd", L.ptr(xd), L.ptr(out), None, n, H + 1, W, C, L.current_stream())
    with pytest.raises(RuntimeError, match="null pointer"):
        L.call("vae_maxpool2_bwd", L.ptr(gd), None, None, L.ptr(out), n, H, W, C, L.current_stream())


@pytest.mark.parametrize("rows,C", [(1, 64), (37, 8), (4096, 64), (100000, 128), (513, 256), (70000, 2048)])
def test_colsum_bf16(L, rows, C):
    """bias gradient of a conv / transposed conv with no normalisation behind it: column sums of its output gradient"""
    g = torch.Generator().manual_seed(rows + C)
    x = torch.randn(rows, C, generator=g).to(torch.bfloat16)
    xd = x.to(DEV)
    nb = L.call("vae_colsum_blocks", rows, C)
    partial = torch.full((nb, C), float("nan"), device=DEV)
    out = torch.full((C,), float("nan"), device=DEV)
    L.call("vae_colsum_bf16", L.ptr(xd), rows, C, L.ptr(partial), L.ptr(out), L.current_stream())
    ref = x.double().sum(0)
    scale = x.double().abs().sum(0).clamp_min(1.0)
    assert float(((out.double().cpu() - ref).abs() / scale).max()) <= 1e-6
    out2 = torch.empty_like(out)
    L.call("vae_colsum_bf16", L.ptr(xd), rows, C, L.ptr(partial), L.ptr(out2), L.current_stream())
    assert torch.equal(out, out2)          # fixed summation order
    with pytest.raises(RuntimeError, match="divisor of 256"):
        L.call("vae_colsum_bf16", L.ptr(xd), rows, 24, L.ptr(partial), L.ptr(out), L.current_stream())


def test_unet_encoder_stage_keeps_the_skip_tensor():
    """block -> (skip tensor, pooled tensor) as the drafts' U-Net encoder does (vae/practice/vae_original.py:142-146):
    the separate pool equals the pool fused into the block, and gradients flow back through both uses"""
    from vae_amd.variants import DoubleConvGN, maxpool2_nhwc
    torch.manual_seed(0)
    blk = DoubleConvGN(3, 64, 0.0).to(DEV).train()
    x = torch.rand(4, 3, 16, 16, device=DEV)
    d0 = blk.forward_nhwc(x)                     # the skip tensor
    p0 = maxpool2_nhwc(d0)
    assert torch.equal(p0, blk.forward_nhwc(x, pool=True))
    w = torch.rand(p0.shape, device=DEV) + 0.5
    (p0.float() * w).sum().backward()
    g_pool_only = [p.grad.clone() for p in blk.parameters()]
    blk.zero_grad()
    (blk.forward_nhwc(x, pool=True).float() * w).sum().backward()
    for a, b in zip(g_pool_only, [p.grad for p in blk.parameters()]):
        assert float((a - b).norm() / b.norm()) <= 1e-2      # the two routes round intermediate gradients to bf16 at different points
    blk.zero_grad()
    d0 = blk.forward_nhwc(x)
    (maxpool2_nhwc(d0).float().sum() + d0.float().sum()).backward()      # both the pooled and the skip path carry gradient
    assert all(torch.isfinite(p.grad).all() and float(p.grad.abs().sum()) > 0 for p in blk.parameters())
