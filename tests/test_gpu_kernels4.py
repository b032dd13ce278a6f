"""GPU parity of the split-K nn.Linear path (fc_mu | fc_logvar forward, fc_up input gradient,
vae/models/vae.py:96-97,116): slabs + fixed-order finish == a torch fp32 matmul of the bf16-rounded
operands, for the plain and the column-split fp32 epilogues, suggested and forced split counts."""
import pytest
import torch

from golden_util import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"
EPI_F32, EPI_F32_SPLIT = 2, 3


@pytest.fixture(scope="module")
def L():
    from vae_amd import _lib
    _lib.load()
    return _lib


def bf(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("M,N,K,ns", [(512, 256, 4096, 0), (512, 128, 4096, 0), (64, 1024, 16384, 0), (7, 64, 256, 2),
                                      (130, 128, 1024, 16), (512, 256, 4096, 1)])
def test_linear_splitk_matches_matmul(L, M, N, K, ns):
    g = torch.Generator().manual_seed(M + N + K)
    a = bf(torch.randn(M, K, generator=g))
    w = bf(torch.randn(N, K, generator=g) / K ** 0.5)
    b = torch.randn(N, generator=g)
    ref = a @ w.t() + b
    s = ns or L.call("vae_linear_splitk", M, N, K)
    assert 1 <= s <= K // 64
    ad, wd, bd = a.to(torch.bfloat16).to(DEV), w.to(torch.bfloat16).to(DEV), b.to(DEV)
    slab = torch.full((s, M, N), float("nan"), device=DEV)
    out = torch.full((M, N), float("nan"), device=DEV)
    L.call("vae_linear_gemm_splitk", L.ptr(ad), L.ptr(wd), L.ptr(bd), L.ptr(out), None, L.ptr(slab), s, M, N, K, EPI_F32,
           L.current_stream())
    assert rel_l2(out.cpu(), ref) <= 1e-5
    # column-split epilogue (mu | logvar), no bias
    o1 = torch.full((M, N // 2), float("nan"), device=DEV)
    o2 = torch.full((M, N // 2), float("nan"), device=DEV)
    L.call("vae_linear_gemm_splitk", L.ptr(ad), L.ptr(wd), None, L.ptr(o1), L.ptr(o2), L.ptr(slab), s, M, N, K,
           EPI_F32_SPLIT, L.current_stream())
    ref0 = a @ w.t()
    assert rel_l2(o1.cpu(), ref0[:, :N // 2]) <= 1e-5 and rel_l2(o2.cpu(), ref0[:, N // 2:]) <= 1e-5
    # the unsplit kernel agrees
    plain = torch.full((M, N), float("nan"), device=DEV)
    L.call("vae_linear_gemm", L.ptr(ad), L.ptr(wd), L.ptr(bd), L.ptr(plain), None, None, None, M, N, K, EPI_F32,
           L.current_stream())
    assert rel_l2(plain.cpu(), out.cpu()) <= 1e-6


def test_linear_splitk_rejects_bad_arguments(L):
    t = torch.zeros(64 * 64, device=DEV)
    with pytest.raises(RuntimeError, match="fp32 epilogues"):
        L.call("vae_linear_gemm_splitk", L.ptr(t), L.ptr(t), None, L.ptr(t), None, L.ptr(t), 1, 64, 64, 64, 0,
               L.current_stream())
    with pytest.raises(RuntimeError, match="out of range"):
        L.call("vae_linear_gemm_splitk", L.ptr(t), L.ptr(t), None, L.ptr(t), None, L.ptr(t), 2, 64, 64, 64, EPI_F32,
               L.current_stream())
    big = torch.zeros(3 * 64 * 64, device=DEV)
    with pytest.raises(RuntimeError, match="empty splits"):     # 4 K-steps cannot be cut into 3 non-empty equal runs
        L.call("vae_linear_gemm_splitk", L.ptr(big), L.ptr(big), None, L.ptr(big), None, L.ptr(big), 3, 64, 64, 256,
               EPI_F32, L.current_stream())
