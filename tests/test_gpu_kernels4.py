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


def test_reparam_kernels_add_up_the_splitk_slabs_themselves(L):
    """fc_mu | fc_logvar (vae/models/vae.py:96-97) -> reparameterize (vae.py:108-111), and fc_up's input gradient ->
    its backward: with out == NULL the split-K GEMM leaves its slabs and the reparameterisation kernel adds them up
    (bias first, slabs in order) -- bit-identical to the GEMM's own finishing step followed by the plain kernel."""
    g = torch.Generator().manual_seed(5)
    M, Lz, K = 96, 128, 4096
    a = bf(torch.randn(M, K, generator=g)).to(torch.bfloat16).to(DEV)
    w = bf(torch.randn(2 * Lz, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(DEV)
    b = torch.randn(2 * Lz, generator=g).to(DEV)
    ns = L.call("vae_linear_splitk", M, 2 * Lz, K)
    assert ns > 1
    n = M * Lz
    grid = L.call("vae_reparam_grid", n)
    s = L.current_stream()

    def outs():
        return [torch.full((M, Lz), float("nan"), device=DEV) for _ in range(3)] + \
               [torch.zeros(M, Lz, device=DEV, dtype=torch.bfloat16), torch.full((grid,), float("nan"), device=DEV)]
    # separate: GEMM + finish, then the plain kernel
    mu1, lv1, z1, zb1, kl1 = outs()
    slab = torch.full((ns, M, 2 * Lz), float("nan"), device=DEV)
    L.call("vae_linear_gemm_splitk", L.ptr(a), L.ptr(w), L.ptr(b), L.ptr(mu1), L.ptr(lv1), L.ptr(slab), ns, M, 2 * Lz, K,
           EPI_F32_SPLIT, s)
    L.call("vae_reparam_kl_fwd", L.ptr(mu1), L.ptr(lv1), None, L.ptr(z1), L.ptr(zb1), None, L.ptr(kl1), n, 77, 3, 1, s)
    # fused: GEMM leaves the slabs
    mu2, lv2, z2, zb2, kl2 = outs()
    slab2 = torch.full((ns, M, 2 * Lz), float("nan"), device=DEV)
    L.call("vae_linear_gemm_splitk", L.ptr(a), L.ptr(w), L.ptr(b), None, None, L.ptr(slab2), ns, M, 2 * Lz, K, EPI_F32_SPLIT, s)
    L.call("vae_reparam_kl_fwd_slabs", L.ptr(slab2), ns, L.ptr(b), L.ptr(mu2), L.ptr(lv2), None, L.ptr(z2), L.ptr(zb2), None,
           L.ptr(kl2), n, Lz, 77, 3, 1, s)
    assert torch.equal(slab, slab2)
    for x1, x2 in ((mu1, mu2), (lv1, lv2), (z1, z2), (zb1, zb2), (kl1, kl2)):
        assert torch.equal(x1, x2)
    assert rel_l2(mu1.cpu(), (a.float() @ w.float().t() + b).cpu()[:, :Lz]) <= 1e-5

    # backward: d z = dpre @ W as slabs
    dpre = bf(torch.randn(M, K, generator=g)).to(torch.bfloat16).to(DEV)
    wu = bf(torch.randn(Lz, K, generator=g) / K ** 0.5).to(torch.bfloat16).to(DEV)
    nz = L.call("vae_linear_splitk", M, Lz, K)
    assert nz > 1
    dz = torch.full((M, Lz), float("nan"), device=DEV)
    sl = torch.full((nz, M, Lz), float("nan"), device=DEV)
    L.call("vae_linear_gemm_splitk", L.ptr(dpre), L.ptr(wu), None, L.ptr(dz), None, L.ptr(sl), nz, M, Lz, K, EPI_F32, s)
    res = []
    for fused in (False, True):
        dmu, dlv = torch.full((M, Lz), float("nan"), device=DEV), torch.full((M, Lz), float("nan"), device=DEV)
        dcat = torch.zeros(M, 2 * Lz, device=DEV, dtype=torch.bfloat16)
        L.call("vae_reparam_kl_bwd_slabs", L.ptr(sl) if fused else L.ptr(dz), nz if fused else 0, L.ptr(mu1), L.ptr(lv1), None,
               None, None, 0.01, L.ptr(dmu), L.ptr(dlv), L.ptr(dcat), n, Lz, 77, 3, 1, s)
        res.append((dmu, dlv, dcat))
    for x1, x2 in zip(*res):
        assert torch.equal(x1, x2)
    with pytest.raises(RuntimeError, match="slabs need dz"):
        L.call("vae_reparam_kl_bwd_slabs", None, 2, L.ptr(mu1), L.ptr(lv1), None, None, None, 0.01, L.ptr(dmu), L.ptr(dlv),
               L.ptr(dcat), n, Lz, 77, 3, 1, s)
