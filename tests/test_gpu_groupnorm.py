"""GroupNorm + ReLU (+ dropout) (+ 2x2 max-pool) kernels against the oracle: forward, the gradient of the conv
output, d gamma / d beta; dropout from an injected mask and from the Philox stream."""
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


def nhwc(t, dtype=torch.bfloat16):
    return t.permute(0, 2, 3, 1).contiguous().to(dtype).to(DEV)


def from_nhwc(t):
    return t.float().cpu().permute(0, 3, 1, 2)


def check_bf16(got, ref, what, rtol=4e-3):
    err = (got - ref).abs()
    tol = ref.abs() * 2.0 ** -8 + 1e-3 * ref.abs().max()
    assert (err <= tol).all(), f"{what}: max err {err.max().item()} (ref max {ref.abs().max().item()})"
    assert rel_l2(got, ref) <= rtol, what


@pytest.mark.parametrize("n,H,W,C,pool,p", [(4, 8, 8, 64, 0, 0.0), (3, 8, 8, 128, 1, 0.0), (2, 16, 16, 64, 1, 0.1),
                                            (5, 4, 4, 256, 0, 0.1), (2, 6, 10, 128, 1, 0.3), (1, 32, 32, 64, 1, 0.1),
                                            (70, 2, 2, 256, 1, 0.0)])
@pytest.mark.parametrize("inject", [False, True])
def test_gn_act_fwd_bwd(L, n, H, W, C, pool, p, inject):
    if inject and p == 0:
        pytest.skip("nothing to inject")
    g = torch.Generator().manual_seed(n + H + C + pool)
    y = bf(torch.randn(n, C, H, W, generator=g) * 1.5 + 0.3 + torch.randn(n, C, 1, 1, generator=g))
    gamma = 0.5 + torch.rand(C, generator=g)
    beta = 0.4 * torch.rand(C, generator=g) - 0.2
    seed, step, stream_id, rank = 77, 2, 1, 3
    if p > 0:
        if inject:
            mask_nhwc = (torch.rand(n, H, W, C, generator=g) >= p).to(torch.uint8)
        else:
            mask_nhwc = torch.from_numpy(O.philox_keep_mask(n * H * W * C, p, seed, stream_id, step, rank)).view(n, H, W, C)
        mask = mask_nhwc.permute(0, 3, 1, 2)
    else:
        mask_nhwc = mask = None
    yr, gr, br = y.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a = O.dropout(torch.relu(O.group_norm(yr, gr, br)), p, True, mask)
    out_ref = F.max_pool2d(a, 2) if pool else a
    gin = bf(torch.randn(out_ref.shape, generator=g))
    (out_ref * gin).sum().backward()

    yd, gd, bd = nhwc(y), gamma.to(DEV), beta.to(DEV)
    md = mask_nhwc.contiguous().to(DEV) if (inject and p > 0) else None
    out = torch.full(tuple(nhwc(out_ref).shape), float("nan"), device=DEV, dtype=torch.bfloat16)
    mr = torch.full((n, 8, 2), float("nan"), device=DEV)
    L.call("vae_gn_act_fwd", L.ptr(yd), L.ptr(gd), L.ptr(bd), L.ptr(out), L.ptr(mr), n, H, W, C, 8, 1e-5, pool, p, L.ptr(md),
           seed, step, stream_id, rank, L.current_stream())
    yg = y.reshape(n, 8, -1)
    assert rel_l2(mr[:, :, 0].cpu(), yg.mean(2)) <= 1e-5
    assert rel_l2(mr[:, :, 1].cpu(), torch.rsqrt(yg.var(2, unbiased=False) + 1e-5)) <= 1e-5
    check_bf16(from_nhwc(out), out_ref.detach(), "gn_act fwd")
    part = torch.full((n, 2, C), float("nan"), device=DEV)
    sums = torch.full((2 * C,), float("nan"), device=DEV)
    dy = torch.full((n, H, W, C), float("nan"), device=DEV, dtype=torch.bfloat16)
    L.call("vae_gn_act_bwd", L.ptr(nhwc(gin)), L.ptr(yd), L.ptr(gd), L.ptr(bd), L.ptr(mr), L.ptr(part), L.ptr(sums), L.ptr(dy),
           n, H, W, C, 8, pool, p, L.ptr(md), seed, step, stream_id, rank, L.current_stream())
    assert rel_l2(sums[:C].cpu(), gr.grad) <= 2e-5
    assert rel_l2(sums[C:].cpu(), br.grad) <= 2e-5
    check_bf16(from_nhwc(dy), yr.grad, "gn_act bwd")


def test_gn_rejects_bad_geometry(L):
    t = torch.zeros(1, 4, 4, 64, device=DEV, dtype=torch.bfloat16)
    f = torch.zeros(64, device=DEV)
    with pytest.raises(RuntimeError, match="multiple of 8 \\* groups"):
        L.call("vae_gn_act_fwd", L.ptr(t), L.ptr(f), L.ptr(f), L.ptr(t), None, 1, 4, 4, 64, 16, 1e-5, 0, 0.0, None, 0, 0, 0, 0,
               L.current_stream())
    with pytest.raises(RuntimeError, match="pooling needs even"):
        L.call("vae_gn_act_fwd", L.ptr(t), L.ptr(f), L.ptr(f), L.ptr(t), None, 1, 3, 4, 64, 8, 1e-5, 1, 0.0, None, 0, 0, 0, 0,
               L.current_stream())
